#!/bin/bash
# config 4: chunks of pairs one main-kernel workgroup draws (P2P_MAIN_SPAN; 3 = all of config 4's), with the turn
# length that goes with it; alternating, three rounds
run() { # label env...
  label=$1; shift
  env "$@" timeout 400 python3 bench.py --workload cfg4 --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --counters none 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %.3f ms per launch, frac %.3f' % (sys.argv[1], j['roofline']['kernel_ms_avg'], j['roofline']['frac']))" "$label"
}
for r in 1 2 3; do
run "span 1, group 96 (round 3's rule)" P2P_MAIN_SPAN=1 P2P_MAIN_GROUP=96
run "span 3, group 96" P2P_MAIN_SPAN=3 P2P_MAIN_GROUP=96
run "span 3, group 48" P2P_MAIN_SPAN=3 P2P_MAIN_GROUP=48
run "span 3, group 24" P2P_MAIN_SPAN=3 P2P_MAIN_GROUP=24
run "span 3, group 32, lead 3" P2P_MAIN_SPAN=3 P2P_MAIN_GROUP=32 P2P_PREFETCH_LEAD=3
run "span 3, group 48, lead 4" P2P_MAIN_SPAN=3 P2P_MAIN_GROUP=48 P2P_PREFETCH_LEAD=4
done
