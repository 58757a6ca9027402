#!/bin/bash
# config 4 with timing-only variants of the plan tables (tools/variants.py build base= halfpx=-DP2P_ABLATE_HALF_PX
# noitems=-DP2P_ABLATE_ITEMS_WINDOW both=... notab=-DP2P_ABLATE_TABLE_WINDOW): what smaller tables could be worth
for round in 1 2; do
  for v in ${AB_VARIANTS:-base halfpx noitems both notab}; do
    P2P_LIB_PATH=$PWD/gpurun_variants/libp2p_$v.so timeout 600 python3 bench.py --workload cfg4 --no-cpu-baseline --no-secondary --counters none --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v cfg4 %.3f ms per launch (kernel sum), %.3f ms per step' % (r['kernel_ms_avg'], j['ms_per_step']))"
  done
done
