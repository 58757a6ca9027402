#!/bin/bash
# the parity fuzzers with the 128-wide kernel's chunk loop forced on (several chunks of 3 / 5 pairs drawn by one
# workgroup), ONE process at a time   bash tools/fuzz_span.sh <tag> [seed]   -> gpurun_out/fuzz_span_<tag>/
TAG=${1:-x}; SEED=${2:-7}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fuzz_span_$TAG
mkdir -p $OUT
cd $ROOT
export P2P_TILE_SHAPE=128
run() { name=$1; shift; timeout ${FZ_TIMEOUT:-900} python3 "$@" > $OUT/$name.log 2>&1; echo "$name: rc $? : $(tail -n 1 $OUT/$name.log | cut -c1-200)"; }
P2P_MAIN_SPAN=2 P2P_PAIRS_PER_BLOCK=3 run parity_span2_of_3   tests/fuzz/fuzz_parity.py --cases ${FZ_PARITY:-600} --seed $((SEED + 10))
P2P_MAIN_SPAN=64 P2P_PAIRS_PER_BLOCK=2 run parity_all_of_2    tests/fuzz/fuzz_parity.py --cases ${FZ_PARITY:-600} --seed $((SEED + 11))
P2P_MAIN_SPAN=3 P2P_PAIRS_PER_BLOCK=5 P2P_MAIN_ORDER=1 P2P_PREFETCH_LEAD=1 run parity_big_span3_list tests/fuzz/fuzz_parity.py --cases ${FZ_BIG:-100} --seed $((SEED + 12)) --mode big
P2P_MAIN_SPAN=4 P2P_PAIRS_PER_BLOCK=1 run parity_real_span4_of_1 tests/fuzz/fuzz_parity.py --cases ${FZ_REAL:-400} --seed $((SEED + 13)) --mode real
P2P_MAIN_SPAN=2 P2P_PAIRS_PER_BLOCK=2 run scramble_span2 tests/fuzz/scramble_tables.py --cases 30 --seed $((SEED + 14))
