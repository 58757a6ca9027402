#!/bin/bash
# Quick SQ passes for bench.py's hot kernel: bash tools/pmc_quick.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for line in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_SMEM SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-preroll --counters none "$@" > $OUT/p$i.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT
