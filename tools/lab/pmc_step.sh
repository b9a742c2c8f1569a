#!/bin/bash
# HBM-side traffic of every kernel of the training step: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2), summarised per kernel into profiles/.   Run on the GPU box:
#   tools/lab/pmc_step.sh r02 [extra bench.py arguments, e.g. --dtype bf16]
set -e
ROOT="$GRAFT_REPO_ROOT"; TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp
export AVVAD_OVERLAP=0
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$ROOT/gpurun_out/pmc_${TAG}_$C" -- python3 "$ROOT/bench.py" --no-extras --steps 2 --warmup 1 "$@" > "$ROOT/gpurun_out/pmc_${TAG}_$C.log" 2>&1
  python3 "$ROOT/tools/pmc_summary.py" "$ROOT"/gpurun_out/pmc_${TAG}_$C/*/*counter_collection.csv $C > "$ROOT/gpurun_out/${TAG}_pmc_$(echo $C | tr A-Z a-z)_per_kernel.csv"
  echo "$C done"
done
