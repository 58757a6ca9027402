#!/bin/bash
# PMC comparison of chunk sizes: bash tools/pmc_chunks.sh   (GPU box, repo root)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_chunks
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ppb in 12 48; do
  i=0
  for line in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_SMEM SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $line --output-format csv -d $OUT/ppb$ppb/p$i -- python3 $ROOT/tools/run_job.py 48 $ppb > $OUT/ppb${ppb}_p$i.log 2>&1
  done
done
python3 - <<PY
import csv, glob, os, re
out = "$OUT"
for ppb in (12, 48):
    vals = {}
    for f in glob.glob(os.path.join(out, "ppb%d" % ppb, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if re.search(r"remap_views_kernel<0,\s*0>", row["Kernel_Name"]):
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print("ppb", ppb)
    for k in sorted(vals):
        v = vals[k]
        print("  %-28s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
PY
