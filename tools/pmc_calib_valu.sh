#!/bin/bash
# What do the SQ VALU counters read on kernels that are VALU-saturated by construction (tools/ubench/valu_rates)?
# Calibrates "VALU busy" for the view kernel.  GPU box, repo root.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_calib
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/u -- $ROOT/tools/ubench/valu_rates > $OUT/u.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ut -- $ROOT/tools/ubench/valu_rates > $OUT/ut.log 2>&1
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/v -- python3 $ROOT/tools/run_job.py 12 0 > $OUT/v.log 2>&1
python3 - <<PY
import csv, glob, os, re
out = "$OUT"
def load(d):
    rows = {}
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
            rows.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return rows
dur = {}
for f in glob.glob(os.path.join(out, "ut", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[re.sub(r"\(.*", "", r["Name"])[:60]] = float(r["AverageNs"])
for d in ("u", "v"):
    for name, c in sorted(load(d).items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if "SQ_ACTIVE_INST_VALU" not in m:
            continue
        line = "%-46s valu_active/busy_cu=%.3f  active/insts=%.2f  wave_cycles/busy_cu=%.2f gui=%.3g busy_cu=%.3g" % (
            name, m["SQ_ACTIVE_INST_VALU"] / max(m["SQ_BUSY_CU_CYCLES"], 1), m["SQ_ACTIVE_INST_VALU"] / max(m["SQ_INSTS_VALU"], 1),
            m["SQ_WAVE_CYCLES"] / max(m["SQ_BUSY_CU_CYCLES"], 1), m["GRBM_GUI_ACTIVE"], m["SQ_BUSY_CU_CYCLES"])
        if name in dur:
            line += "  dur_us=%.1f" % (dur[name] / 1e3)
        print(line)
PY
