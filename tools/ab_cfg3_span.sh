#!/bin/bash
# config 3, all 64 panoramas on one GPU (128-wide tiles, 32 chunks of 24 pairs) and its share of 16: chunks of pairs per
# main-kernel workgroup (P2P_MAIN_SPAN)
run() { # label panos env...
  label=$1; n=$2; shift; shift
  env "$@" timeout 400 python3 bench.py --workload cfg3 --scaling weak --panos-per-gpu $n --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --counters none 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %.3f ms per launch, frac %.3f' % (sys.argv[1], j['roofline']['kernel_ms_avg'], j['roofline']['frac']))" "$label"
}
for r in 1 2; do
for n in 64 16; do
run "$n panos, tile 128, span 1" $n P2P_TILE_SHAPE=128 P2P_MAIN_SPAN=1
run "$n panos, tile 128, span 2" $n P2P_TILE_SHAPE=128 P2P_MAIN_SPAN=2
run "$n panos, tile 128, span 4" $n P2P_TILE_SHAPE=128 P2P_MAIN_SPAN=4
done
done
