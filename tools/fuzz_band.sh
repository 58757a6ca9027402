#!/bin/bash
# tools/fuzz_round.sh's parity runs once more with source-band tiles FORCED wherever they apply (P2P_BAND=1; in the band
# shape unless P2P_TILE_SHAPE names another), scrambled
# band tables included, and the first-launch knobs off; one process at a time.   bash tools/fuzz_band.sh <tag> [seed]
TAG=${1:-x}; SEED=${2:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fuzz_band_$TAG
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; timeout ${FZ_TIMEOUT:-900} "$@" > $OUT/$name.log 2>&1; echo "$name: rc $? : $(tail -n 1 $OUT/$name.log | cut -c1-200)"; }
export P2P_BAND=1
run band_parity      python3 tests/fuzz/fuzz_parity.py --cases ${FZ_PARITY:-400} --seed $((SEED + 20))
run band_parity_big  python3 tests/fuzz/fuzz_parity.py --cases ${FZ_BIG:-60} --seed $((SEED + 21)) --mode big
run band_parity_real python3 tests/fuzz/fuzz_parity.py --cases ${FZ_REAL:-200} --seed $((SEED + 22)) --mode real
run band_w64         env P2P_TILE_SHAPE=64 python3 tests/fuzz/fuzz_parity.py --cases ${FZ_PARITY:-400} --seed $((SEED + 28))
run band_w128        env P2P_TILE_SHAPE=128 python3 tests/fuzz/fuzz_parity.py --cases ${FZ_PARITY:-400} --seed $((SEED + 23))
run band_cells_4x32  env P2P_BAND_BH=4 P2P_BAND_CW=32 python3 tests/fuzz/fuzz_parity.py --cases ${FZ_PARITY:-400} --seed $((SEED + 24))
run band_scramble    python3 tests/fuzz/scramble_tables.py --cases 30 --seed $((SEED + 25))
run band_oneshot     python3 tests/fuzz/fuzz_oneshot.py --cases ${FZ_ONESHOT:-100} --threads 4 --seed $((SEED + 26))
unset P2P_BAND
run band_race        python3 tests/fuzz/band_race.py 30 300
run late_lists       env P2P_EARLY_MAIN=0 P2P_DEFER_LISTS=0 python3 tests/fuzz/fuzz_parity.py --cases ${FZ_PARITY:-400} --seed $((SEED + 27))
