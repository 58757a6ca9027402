#!/bin/bash
# First GPU call of a round: the evidence the previous round could not collect at its end (its gpurun closed after a
# test script was started with nine thousand threads: tests/fuzz/README.md), one process after the other, each under
# its own timeout.   GPU box, repo root:  bash tools/round_start_check.sh <tag>   -> gpurun_out/start_<tag>/
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/start_$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu: rc $? : $(tail -n 1 $OUT/pytest_gpu.log)"
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke: rc $? : $(tail -n 1 $OUT/smoke.log)"
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench: rc $? : $(cut -c1-160 $OUT/bench.json)"
timeout 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmdline.json 2> $OUT/bench_driver.err; echo "bench (driver's command line): rc $?"
# the two fuzzers that have not run on the builds with the work lists -- with their INTENDED arguments:
# fuzz_remap.py [cases] [seed];  fuzz_oneshot.py [calls per thread] [THREADS <= 32] [seed]
timeout 300 python3 tests/fuzz/fuzz_remap.py 600 5 > $OUT/fuzz_remap.log 2>&1; echo "fuzz_remap: rc $? : $(tail -n 1 $OUT/fuzz_remap.log)"
timeout 300 python3 tests/fuzz/fuzz_oneshot.py 150 4 7 > $OUT/fuzz_oneshot.log 2>&1; echo "fuzz_oneshot: rc $? : $(tail -n 1 $OUT/fuzz_oneshot.log)"
