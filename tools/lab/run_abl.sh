#!/bin/bash
# run the conv-forward microbench against each ablated build of the engine (tools/lab/libs/lib_<ABL>.so)
cd "$(dirname "$0")/../.."
echo "== baseline"; timeout -k 10 120 python tools/mb_abl.py 2>&1 | grep conv
for l in tools/lab/libs/lib_*.so; do
  echo "== $l"; AVVAD_LIB=$PWD/$l timeout -k 10 120 python tools/mb_abl.py 2>&1 | grep conv
done
