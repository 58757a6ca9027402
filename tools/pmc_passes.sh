#!/bin/bash
# Collect rocprofv3 PMC passes for bench.py's hot kernel.  Usage (on the GPU box, from the repo root):
#   bash tools/pmc_passes.sh <tag> [bench args...]
# Each pass is its own run with --pmc only (never combined with tracing), outputs under gpurun_out/pmc_<tag>/.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-preroll --counters none "$@" > $OUT/p$i.log 2>&1
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY
SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM
TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE GRBM_COUNT
PASSES
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
