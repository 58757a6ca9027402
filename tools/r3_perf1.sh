#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
B="python3 bench.py --no-cpu-baseline --no-secondary --counters none --steps 1500 --warmup 300"
show() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-22s %9.2f us/launch %8.1f Gpix/s frac %.3f' % (sys.argv[1], r['kernel_ms_avg']*1e3, j['value']/1e3, r['frac']))" "$1"; }
for rep in 1 2; do
$B 2>/dev/null | show "shipped"
P2P_LIB_PATH=$ROOT/gpurun_variants/lib_nowrap.so $B 2>/dev/null | show "no wrap test (ablation)"
P2P_LIB_PATH=$ROOT/gpurun_variants/lib_noconf.so $B 2>/dev/null | show "no bank conflicts (abl.)"
done
bash tools/r3_cfg4c.sh
python3 tests/fuzz/parity_report.py > gpurun_out/parity_report.json 2> gpurun_out/parity_report.err; tail -60 gpurun_out/parity_report.json
