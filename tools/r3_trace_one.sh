#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_trace_one
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ppb in 1 4; do
export P2P_GATHER_PPB=$ppb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$ppb -- python3 $ROOT/tools/probe_job.py 8192 4096 800 800 90 0,90,180,270 90 > $OUT/p$ppb.log 2>&1
grep "us per" $OUT/p$ppb.log
for f in $(find $OUT/t$ppb -name "*kernel_stats.csv"); do python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   %-40s calls %6s avg %10.1f ns" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])))
PY
done
done
