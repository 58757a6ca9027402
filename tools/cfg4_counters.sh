#!/bin/bash
# counters of config 4 AS IT SHIPS (72 yaws x 5 pitches in one job: 128-wide tiles, list order, table prefetch), main and
# gather kernel, one rocprofv3 --pmc pass per counter group:   bash tools/cfg4_counters.sh  -> gpurun_out/cfg4_counters/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cfg4_counters
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 30,60,90,120,150 3 > $OUT/$name.log 2>&1
  python3 - $OUT/$name <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "gather" if "gather_kernel" in r["Kernel_Name"] else ("main" if "remap_views_kernel" in r["Kernel_Name"] else None)
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: "%.5g" % (sum(v) / len(v)) for c, v in sorted(acc[k].items())})
PY
}
{
run f FETCH_SIZE
run w WRITE_SIZE
run t TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run s SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
run l SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM
} 2>&1 | tee $OUT/summary.txt
