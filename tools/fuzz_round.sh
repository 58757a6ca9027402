#!/bin/bash
# the randomised parity runs of tests/fuzz/ on the current build, ONE process at a time, each under its own timeout
# (named options, hard caps: tests/fuzz/_args.py).   bash tools/fuzz_round.sh <tag> [seed]   -> gpurun_out/fuzz_<tag>/
TAG=${1:-x}; SEED=${2:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fuzz_$TAG
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; timeout ${FZ_TIMEOUT:-900} python3 "$@" > $OUT/$name.log 2>&1; echo "$name: rc $? : $(tail -n 1 $OUT/$name.log | cut -c1-200)"; }
run parity      tests/fuzz/fuzz_parity.py --cases ${FZ_PARITY:-400} --seed $((SEED + 10))
run parity_big  tests/fuzz/fuzz_parity.py --cases ${FZ_BIG:-60} --seed $((SEED + 11)) --mode big
run parity_real tests/fuzz/fuzz_parity.py --cases ${FZ_REAL:-200} --seed $((SEED + 12)) --mode real
run remap       tests/fuzz/fuzz_remap.py --cases ${FZ_REMAP:-800} --seed $((SEED + 13))
run fused       tests/fuzz/fuzz_fused.py --cases ${FZ_FUSED:-120} --seed $((SEED + 14))
run float       tests/fuzz/fuzz_float.py --cases ${FZ_FLOAT:-120} --seed $((SEED + 15))
run oneshot     tests/fuzz/fuzz_oneshot.py --cases ${FZ_ONESHOT:-100} --threads 4 --seed $((SEED + 16))
run scramble    tests/fuzz/scramble_tables.py --cases 30 --seed $((SEED + 17))
run rows        tests/fuzz/fuzz_parity.py --cases ${FZ_REAL:-200} --seed $((SEED + 18)) --rows
run exact       tests/fuzz/fuzz_exact.py --cases ${FZ_EXACT:-200} --seed $((SEED + 19))
