#!/bin/bash
# PMC passes over the single-layer forward kernel (run on the GPU box): tools/lab/pmc_layer.sh <tag>   (env AVVAD_WN_FLAT etc. pass through)
set -e
ROOT="$GRAFT_REPO_ROOT"; TAG="$1"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA"
P3="TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES"
P4="TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_READ_TAGCONFLICT_STALL_CYCLES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d "$ROOT/gpurun_out/pmc_layer_${TAG}_$i" -- python3 "$ROOT/tools/lab/one_layer.py" 256 16000 64 3 > "$ROOT/gpurun_out/pmc_layer_${TAG}_$i.log" 2>&1
done
echo pmc done
