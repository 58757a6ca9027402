#!/bin/bash
# round-3 iteration check: GPU parity tests, then the bench lines / probes that moved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_check
mkdir -p $OUT
cd $ROOT
timeout 1500 python3 -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
python3 tools/probe_default_cli.py > $OUT/cli.log 2>&1; cat $OUT/cli.log
python3 bench.py --no-cpu-baseline --counters none > $OUT/cfg2.json 2> $OUT/cfg2.err; python3 -c "
import json,sys
for f in sys.argv[1:]:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j['roofline']
        print('%-30s %9.1f us/launch %8.1f Gpix/s frac %.3f' % (f.split('/')[-1], r['kernel_ms_avg']*1e3, j['value']/1e3, r['frac']))
    except Exception as e: print(f, 'FAILED', e)
" $OUT/cfg2.json
python3 bench.py --workload cfg4 --steps 20 --warmup 5 --no-cpu-baseline --counters none > $OUT/cfg4.json 2> $OUT/cfg4.err
python3 bench.py --workload cfg5 --steps 100 --warmup 20 --no-cpu-baseline --counters none > $OUT/cfg5.json 2> $OUT/cfg5.err
python3 -c "
import json,sys
for f in sys.argv[1:]:
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j['roofline']
        print('%-30s %9.1f us/launch %8.1f Gpix/s frac %.3f' % (f.split('/')[-1], r['kernel_ms_avg']*1e3, j['value']/1e3, r['frac']))
    except Exception as e: print(f, 'FAILED', e)
" $OUT/cfg4.json $OUT/cfg5.json
