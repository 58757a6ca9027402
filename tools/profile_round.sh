#!/bin/bash
# One-shot profile collection for a round: bash tools/profile_round.sh r01   (on the GPU box, repo root)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --counters none > $OUT/bench_under_trace.json 2> $OUT/trace.log
# the driver's own command line, under the trace too (20 timed launches after the disclosed pre-roll)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --counters none > $OUT/bench_driver_under_trace.json 2> $OUT/trace_driver.log
# the other BASELINE configs and the reference CLI's default view set: per-kernel split (main / gather / plan)
for wl in cfg4 cfg5 cli; do
  case $wl in cfg4) st="--steps 8 --warmup 3";; cfg5) st="--steps 100 --warmup 20";; *) st="--steps 1000 --warmup 200";; esac
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -- python3 $ROOT/bench.py --workload $wl $st --no-cpu-baseline --no-secondary --counters none > $OUT/bench_${wl}_under_trace.json 2> $OUT/trace_$wl.log
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-preroll --counters none > /dev/null 2> $OUT/pmc_$c.log
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$c -- $ROOT/tools/ubench/fetch_calib > /dev/null 2> $OUT/calib_$c.log
done
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/pmc_ea -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-preroll --counters none > /dev/null 2> $OUT/pmc_ea.log
# the SQ pass bench.py's issue floor comes from (roofline.issue_floor_us): SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / the clock
# the kernel holds (SQ_BUSY_CU_CYCLES / 256 CUs / the counted dispatches' own duration)
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-preroll --counters none > /dev/null 2> $OUT/pmc_sq.log
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/pmc_tcc -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-preroll --counters none > /dev/null 2> $OUT/pmc_tcc.log
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python3 - <<PY
import csv, glob, json, os, re
out = "$OUT"
def counters(d, key):
    # key "remap_views_kernel" = the main kernel only (remap_views_rest_kernel and plan_kernel are others)
    pat = re.compile(r"remap_views_kernel[<(]") if key == "remap_views_kernel" else re.compile(re.escape(key))
    vals = {}
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pat.search(row["Kernel_Name"]):
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}
res = {"views": {}, "calib": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    res["views"].update(counters("pmc_" + c, "remap_views_kernel"))
    res["calib"]["pieces_" + c] = counters("calib_" + c, "pieces").get(c)
    res["calib"]["stream16_" + c] = counters("calib_" + c, "stream16").get(c)
res["views"].update(counters("pmc_ea", "remap_views_kernel"))
res["views"].update(counters("pmc_tcc", "remap_views_kernel"))
res["views"].update(counters("pmc_sq", "remap_views_kernel"))
# the counted dispatches' own duration (the clock is derived from it) and the issue floor
dur = {}
for f in glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if re.search(r"remap_views_kernel[<(]", row["Kernel_Name"]):
            try:
                dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
v = res["views"]
if dur and v.get("SQ_BUSY_CU_CYCLES") and v.get("SQ_INSTS_VALU"):
    ns = sum(dur.values()) / len(dur)
    clock = v["SQ_BUSY_CU_CYCLES"] / 256.0 / (ns * 1e-9)
    res["issue_floor"] = {"sq_pass_kernel_us": ns / 1e3, "clock_ghz": clock / 1e9,
                          "issue_floor_us": v["SQ_INSTS_VALU"] * 4.0 / 1024.0 / clock * 1e6,
                          "valu_lane_insts_per_output_px": v["SQ_INSTS_VALU"] * 64.0 / (36 * 1920 * 1080),
                          "how": "SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / (SQ_BUSY_CU_CYCLES / 256 CUs / the counted dispatches' duration)"}
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cat $(find $OUT/trace -name "*kernel_stats.csv" | head -1) | cut -c1-160
cat $(find $OUT/trace_driver -name "*kernel_stats.csv" | head -1) | cut -c1-160
for wl in cfg4 cfg5 cli; do cat $(find $OUT/trace_$wl -name "*kernel_stats.csv" | head -1) | cut -c1-160; done
cat $OUT/bench.json
