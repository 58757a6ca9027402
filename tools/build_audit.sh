#!/bin/bash
# The audit build of the library (-DP2P_AUDIT, csrc/p2p_audit.h): every range check of the view kernels records its
# first violation, every plan pool is poisoned before the plan pass, every p2p_job_run waits and reads the record.
#   bash tools/build_audit.sh            -> gpurun_variants/libp2p_hip_audit.so
#   P2P_LIB_PATH=gpurun_variants/libp2p_hip_audit.so python tests/fuzz/stress_audit.py ...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_variants
cd $ROOT && python3 - <<PY
import importlib, sys
sys.path.insert(0, "$ROOT")
b = importlib.import_module("360-to-planer-images_amd._build")
print(b.build(force=True, out="$ROOT/gpurun_variants/libp2p_hip_audit.so", extra_flags=["-DP2P_AUDIT"]))
PY
