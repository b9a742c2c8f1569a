#!/bin/bash
# Build libavvad_hip.so for gfx950 (MI355X).  Usage: csrc/build.sh [extra hipcc flags]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/_obj"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$HERE/../../include -I$HERE $*"
pids=()
for f in gemm trunk lstm misc wavenet mcb stft; do
  ( hipcc $FLAGS -c "$HERE/$f.hip" -o "$HERE/_obj/$f.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libavvad_hip.so" "$HERE"/_obj/{gemm,trunk,lstm,misc,wavenet,mcb,stft}.o
echo "built $OUT/libavvad_hip.so"
