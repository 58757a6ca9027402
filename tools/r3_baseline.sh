#!/bin/bash
# round-3 starting point: per-kernel split of the CLI default view set and of config 4, bench lines of cfg 2 / 4 / 5
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_base
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cli -- python3 $ROOT/tools/probe_default_cli.py > $OUT/cli.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg4 -- python3 $ROOT/bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --counters none > $OUT/cfg4_trace.json 2> $OUT/cfg4_trace.err
cd $ROOT
python3 bench.py --no-cpu-baseline --no-secondary --counters none > $OUT/cfg2.json 2> $OUT/cfg2.err
python3 bench.py --workload cfg4 --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --counters none > $OUT/cfg4.json 2> $OUT/cfg4.err
python3 bench.py --workload cfg5 --steps 200 --warmup 40 --no-cpu-baseline --no-secondary --counters none > $OUT/cfg5.json 2> $OUT/cfg5.err
python3 tools/probe_default_cli.py > $OUT/cli_plain.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
tail -3 $OUT/cfg2.json | cut -c1-600
cat $OUT/cli_plain.log
