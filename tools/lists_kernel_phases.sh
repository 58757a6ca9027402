#!/bin/bash
# Where the 25 us of main_lists_kernel go (csrc/p2p_lists.hip): timing-only builds that return after phase A (counts), B
# (band offsets), C (scatter), D (running sums + cuts) or before E (table) -- -DP2P_LISTS_STOP_AFTER=1..5, built with
# tools/variants.py into gpurun_variants/libp2p_s<N>.so next to libp2p_full.so -- each under rocprofv3 --kernel-trace.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lists_phases
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for so in $ROOT/gpurun_variants/libp2p_s*.so $ROOT/gpurun_variants/libp2p_full.so; do
  name=$(basename $so .so)
  export P2P_LIB_PATH=$so
  rm -rf $OUT/$name
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $ROOT/tools/cold_timeline.py 4 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  echo "$name: $(grep main_lists_kernel $f | cut -d, -f2-4,6-7)   (calls, total ns, average ns, min, max)"
  rm -rf $OUT/$name
done
