#!/bin/bash
# SQ counter passes over bench.py's hot kernel (each pass its own run, --pmc only).  On the GPU box, repo root:
#   bash tools/pmc_sq.sh <tag> [bench args...]      (P2P_LIB_PATH selects a variant build)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-preroll --counters none "$@" > $OUT/p$i.log 2>&1
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY
SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_I8
PASSES
python3 $ROOT/tools/pmc_summary.py $OUT remap_views_kernel > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
