#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
B="python3 bench.py --no-cpu-baseline --no-secondary --counters none --panos-per-gpu 8 --steps 200 --warmup 40"
show() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-40s %9.2f us/launch %8.1f Gpix/s frac %.3f' % (sys.argv[1], r['kernel_ms_avg']*1e3, j['value']/1e3, r['frac']))" "$1"; }
$B 2>/dev/null | show "8 panos: chunk = panorama, outermost"
P2P_CHUNK_OUTER=0 $B 2>/dev/null | show "8 panos: chunk = panorama, inner"
P2P_MAX_PAIRS_PER_BLOCK=16 $B 2>/dev/null | show "8 panos: 16 pairs, outermost"
P2P_MAX_PAIRS_PER_BLOCK=16 P2P_CHUNK_OUTER=0 $B 2>/dev/null | show "8 panos: 16 pairs, inner"
P2P_MAX_PAIRS_PER_BLOCK=24 $B 2>/dev/null | show "8 panos: 24 pairs, outermost"
