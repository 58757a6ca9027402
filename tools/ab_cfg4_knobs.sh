#!/bin/bash
# config 4 (72 yaws x 5 pitches, 128-wide tiles): the list order's turn length, the table-prefetch lead and the pairs per
# workgroup, one knob at a time against the library's own rule
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { # label env...
  label=$1; shift
  env "$@" timeout 400 python3 bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --counters none 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.3f ms per launch, frac %.3f' % ('$label', j['roofline']['kernel_ms_avg'], j['roofline']['frac']))"
}
run "library's rule" P2P_PLAN_CACHE=1
run "main_group 48" P2P_MAIN_GROUP=48
run "main_group 192" P2P_MAIN_GROUP=192
run "prefetch_lead 0" P2P_PREFETCH_LEAD=0
run "prefetch_lead 1" P2P_PREFETCH_LEAD=1
run "prefetch_lead 4" P2P_PREFETCH_LEAD=4
run "pairs_per_block 24" P2P_PAIRS_PER_BLOCK=24
run "pairs_per_block 18" P2P_PAIRS_PER_BLOCK=18
run "pairs_per_block 48" P2P_MAX_PAIRS_PER_BLOCK=64 P2P_PAIRS_PER_BLOCK=48
run "tile shape 64" P2P_TILE_SHAPE=64
run "library's rule" P2P_PLAN_CACHE=1
