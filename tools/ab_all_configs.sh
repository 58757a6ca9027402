#!/bin/bash
# every library in gpurun_variants/ named in AB_VARIANTS through configs 2, 5, 3 (share of 8, all 64), 4 and the CLI set,
# alternating, AB_ROUNDS times
run() { # variant workload steps extra...
  v=$1; w=$2; st=$3; shift; shift; shift
  P2P_LIB_PATH=$PWD/gpurun_variants/libp2p_$v.so timeout 400 python3 bench.py --workload $w --steps $st --warmup $((st/4)) --no-cpu-baseline --no-secondary --counters none "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s %-40s %9.1f us per launch, frac %.3f' % (sys.argv[1], sys.argv[2], 1e3*j['roofline']['kernel_ms_avg'], j['roofline']['frac']))" "$v" "$w $*"
}
for r in $(seq 1 ${AB_ROUNDS:-2}); do
  for v in $AB_VARIANTS; do run $v cfg2 1000; done
  for v in $AB_VARIANTS; do run $v cfg5 200; done
  for v in $AB_VARIANTS; do run $v cfg3 300 --scaling weak --panos-per-gpu 8; done
  for v in $AB_VARIANTS; do run $v cfg3 30 --scaling weak --panos-per-gpu 64; done
  for v in $AB_VARIANTS; do run $v cfg4 30; done
  for v in $AB_VARIANTS; do run $v cli 1000; done
done
