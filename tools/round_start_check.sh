#!/bin/bash
# First GPU call of a round: parity tests, smoke, the two bench lines -- one process after the other, each under its
# own timeout, and NOTHING else in the call (no fuzzers, no stress: a lost box loses these logs with it).  Copy the pulled
# logs into profiles/ and commit them before any other GPU work.   GPU box, repo root:  bash tools/round_start_check.sh <tag>   -> gpurun_out/start_<tag>/
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/start_$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu: rc $? : $(tail -n 1 $OUT/pytest_gpu.log)"
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke: rc $? : $(tail -n 1 $OUT/smoke.log)"
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench: rc $? : $(cut -c1-160 $OUT/bench.json)"
timeout 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmdline.json 2> $OUT/bench_driver.err; echo "bench (driver's command line): rc $?"
