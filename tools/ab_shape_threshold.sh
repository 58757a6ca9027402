for r in 1 2; do for n in ${PANOS:-4 8 16}; do for sh in 64 128; do
P2P_TILE_SHAPE=$sh timeout 300 python3 bench.py --workload cfg3 --scaling weak --panos-per-gpu $n --steps 200 --warmup 50 --no-cpu-baseline --no-secondary --counters none 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('panos %s tile %s: %.1f us per launch, frac %.3f' % (sys.argv[1], sys.argv[2], 1e3*j['roofline']['kernel_ms_avg'], j['roofline']['frac']))" $n $sh
done; done; done
