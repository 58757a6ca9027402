for v in ${AB_VARIANTS:-nonw nw nonw nw}; do
  P2P_LIB_PATH=$PWD/gpurun_variants/libp2p_$v.so timeout 600 python3 bench.py --no-cpu-baseline --counters none 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v cfg2 %.1f us' % (r['kernel_ms_avg']*1e3), {k:round(v.get('ms_per_launch',0),4) for k,v in j['secondary'].items()})"
done
