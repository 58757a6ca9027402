#!/bin/bash
# counters of the main view kernel on config 4's geometry: one pitch vs five in one job
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cfg4_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name pitches counters...
  name=$1; pitches=$2; shift 2
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 $pitches 3 > $OUT/$name.log 2>&1
  python3 - $OUT/$name $name <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "remap_views_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
}
for p in 90 90,90,90,90,90; do
  tag=$(echo $p | tr ',' '_')
  run f_$tag $p FETCH_SIZE
  run w_$tag $p WRITE_SIZE
  run t_$tag $p TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
  run s_$tag $p SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
  run g_$tag $p GRBM_GUI_ACTIVE
done
