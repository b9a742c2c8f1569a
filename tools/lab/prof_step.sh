#!/bin/bash
# kernel-trace stats of the training step without stream overlap (per-kernel durations are then not inflated by sharing):
# tools/lab/prof_step.sh <out-name>   (on the GPU box)
set -e
ROOT="$GRAFT_REPO_ROOT"; NAME="$1"
cd /tmp && export TMPDIR=/tmp
export AVVAD_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$NAME" -- python3 "$ROOT/bench.py" --no-extras --steps 10 --warmup 3 > "$ROOT/gpurun_out/prof_$NAME.log" 2>&1
cp "$ROOT"/gpurun_out/prof_$NAME/*/*kernel_stats.csv "$ROOT/gpurun_out/${NAME}_kernel_stats.csv"
echo profiled
