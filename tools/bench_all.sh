#!/bin/bash
# The bench lines of every workload / mode DESIGN.md 6.1 quotes: bash tools/bench_all.sh <tag>   (GPU box, repo root)
# -> gpurun_out/bench_<tag>/<name>_bench.json   (copy the ones to be judged into profiles/<tag>_*)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bench_$TAG
mkdir -p $OUT
cd $ROOT
B="python3 bench.py --no-cpu-baseline --no-secondary --counters none"
$B --workload cfg1                                   > $OUT/cfg1_bench.json 2> $OUT/cfg1.err
python3 bench.py                                     > $OUT/cfg2_bench.json 2> $OUT/cfg2.err
python3 bench.py --steps 20 --warmup 5               > $OUT/cfg2_bench_driver_cmdline.json 2> $OUT/cfg2d.err
$B --kind N                                          > $OUT/noise_bench.json 2> $OUT/noise.err
$B --maps caller                                     > $OUT/caller_bench.json 2> $OUT/caller.err
$B --panos-per-gpu 8 --steps 300 --warmup 50         > $OUT/8p_bench.json 2> $OUT/8p.err
$B --workload cfg3 --steps 40 --warmup 10            > $OUT/cfg3_64panos_1gpu_bench.json 2> $OUT/cfg3.err
$B --workload cfg4 --steps 30 --warmup 8             > $OUT/cfg4_bench.json 2> $OUT/cfg4.err
$B --workload cfg5 --steps 200 --warmup 40           > $OUT/cfg5_bench.json 2> $OUT/cfg5.err
$B --workload cfg5 --pixel-path f16 --steps 200 --warmup 40 > $OUT/cfg5_f16_bench.json 2> $OUT/cfg5h.err
$B --workload cfg5 --pixel-path f32 --steps 200 --warmup 40 > $OUT/cfg5_f32_bench.json 2> $OUT/cfg5f.err
$B --scaling strong --workload cfg2                  > $OUT/cfg2_views_sharded_1rank_bench.json 2> $OUT/cfg2s.err
$B --workload cli --steps 1000 --warmup 200          > $OUT/cli_default_bench.json 2> $OUT/cli.err
python3 - <<PY
import glob, json, os
for f in sorted(glob.glob("$OUT/*_bench*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-40s %9.1f us/launch  %7.1f Gpix/s  frac %.3f" % (os.path.basename(f), r["kernel_ms_avg"] * 1e3, j["value"] / 1e3, r["frac"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
