cd $GRAFT_REPO_ROOT
P2P_LIB_PATH=gpurun_variants/libp2p_hip_audit.so timeout 900 python3 -m pytest tests/test_gpu_views_exact.py tests/test_gpu_fullsize.py tests/test_gpu_views_fused.py -x -q 2>&1 | tail -2
P2P_LIB_PATH=gpurun_variants/libp2p_hip_audit.so timeout 300 python3 tests/fuzz/scramble_tables.py 40 6 2>&1 | tail -1
