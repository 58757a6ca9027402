#!/bin/bash
# the last entries of every XCD's main list drawn by several workgroups of a part of the pairs each (P2P_MAIN_TAIL entries
# per XCD, 0 = off; P2P_MAIN_TAIL_PARTS 2..4): parity first, then config 2
for t in "0 2" "44 2" "500 2" "30 3" "500 4"; do set -- $t
  P2P_MAIN_TAIL=$1 P2P_MAIN_TAIL_PARTS=$2 timeout 900 python3 -m pytest tests/test_gpu_views_exact.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not cfg4 and not cfg3" 2>&1 | tail -1 | sed "s/^/P2P_MAIN_TAIL=$1 PARTS=$2: /"
done
run() { # tail parts
  P2P_MAIN_TAIL=$1 P2P_MAIN_TAIL_PARTS=$2 timeout 400 python3 bench.py --steps 1000 --warmup 250 --no-cpu-baseline --no-secondary --counters none 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 tail %-4s parts %s %9.1f us per launch, frac %.3f' % (sys.argv[1], sys.argv[2], 1e3*j['roofline']['kernel_ms_avg'], j['roofline']['frac']))" $1 $2
}
for r in 1 2 3; do run 0 2; run 30 2; run 44 2; run 15 3; run 30 3; run 44 3; run 15 4; run 30 4; run 44 4; done
