#!/bin/bash
# Instruction-fetch and wait-reason counters of bench.py's hot kernel (counters only, each pass its own run).
#   bash tools/pmc_ifetch.sh <tag> [bench args...]      (P2P_LIB_PATH selects a variant build)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ[C]*_[A-Z0-9_]*" | sort -u > $OUT/counters_available.txt
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-preroll --counters none "$@" > $OUT/p$i.log 2>&1
done <<'PASSES'
SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM
SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES
PASSES
python3 $ROOT/tools/pmc_summary.py $OUT remap_views_kernel > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
