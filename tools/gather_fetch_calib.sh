#!/bin/bash
# bash tools/gather_fetch_calib.sh  -> gpurun_out/gather_calib/summary.txt (see tools/gather_fetch_calib.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gather_calib
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1 4; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f$n -- python3 $ROOT/tools/gather_fetch_calib.py $n > $OUT/f$n.log 2>&1
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/t$n -- python3 $ROOT/tools/gather_fetch_calib.py $n > $OUT/t$n.log 2>&1
  grep "^views" $OUT/f$n.log
  python3 $ROOT/tools/pmc_summary.py $OUT/f$n remap_views_gather_kernel | sed "s/^/  /"
  python3 $ROOT/tools/pmc_summary.py $OUT/t$n remap_views_gather_kernel | sed "s/^/  /"
done 2>&1 | tee $OUT/summary.txt
