#!/bin/bash
# iteration check of a round: GPU parity tests (log kept), then the bench line with its secondary configs.
#   bash tools/gpu_check.sh <tag>   -> gpurun_out/check_<tag>/
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/check_$TAG
mkdir -p $OUT
cd $ROOT
timeout 2400 python3 -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1
echo "pytest -m gpu: rc $?"; tail -12 $OUT/pytest_gpu.log
timeout 600 python3 bench.py --no-cpu-baseline --counters none > $OUT/cfg2.json 2> $OUT/cfg2.err; tail -3 $OUT/cfg2.err
python3 - $OUT/cfg2.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j['roofline']
print('cfg2 %9.1f us/launch %8.1f Gpix/s frac %.3f' % (r['kernel_ms_avg']*1e3, j['value']/1e3, r['frac']))
print('cold', json.dumps(j.get('cold')))
for k,v in (j.get('secondary') or {}).items():
    print('%-26s' % k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_launch','Gpix_s','frac_of_hbm_peak','plan_ms','yaw_tables_ms','error')})
PY
