#!/bin/bash
# Sample the GPU clock and power while bench.py runs (is the kernel clock- or power-limited?).
# Usage on the GPU box: bash tools/clock_probe.sh [bench args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python3 bench.py --no-cpu-baseline --no-secondary --steps 40000 --warmup 50 "$@" > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
sleep 5
for i in $(seq 1 16); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' '
  echo
  sleep 0.25
done
wait $BP
cut -c1-200 gpurun_out/clock_bench.json
