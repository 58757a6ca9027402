#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_trace_cli
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/probe_default_cli.py > $OUT/cli.log 2>&1
cat $OUT/cli.log
for f in $(find $OUT -name "*kernel_stats.csv"); do cut -c1-60,200-400 $f | head -8; done
for ppb in 1 2 4; do echo "P2P_GATHER_PPB=$ppb"; P2P_GATHER_PPB=$ppb python3 $ROOT/tools/probe_default_cli.py 2>&1 | tail -5; done
