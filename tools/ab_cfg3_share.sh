#!/bin/bash
# config 3's share of one GPU (8 resident panoramas x 36 views): pairs per workgroup (12 = exactly one panorama's yaws
# per workgroup) and list / grid order
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for cfg in "0 0" "12 0" "24 0" "12 2" "0 2"; do
  set -- $cfg
  P2P_PAIRS_PER_BLOCK=$1 P2P_MAIN_ORDER=$([ $2 = 0 ] && echo -1 || echo $2) timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --counters none --panos-per-gpu 8 --steps 300 --warmup 50 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ppb $1 main_order $2: %.1f us per launch, frac %.3f' % (j['roofline']['kernel_ms_avg']*1e3, j['roofline']['frac']))"
done; done
