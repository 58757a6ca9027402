#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_views_exact.py tests/test_gpu_api.py tests/test_gpu_float_path.py -m gpu -q -x --timeout 600 2>&1 | tail -2
python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 30,60,90,120,150 10 2>&1 | grep -E "us per"
python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 90,90,90,90,90 8 2>&1 | grep -E "us per"
python3 tools/probe_job.py 8192 4096 1920 1080 90 0:360:30 60,90,120 2000 2>&1 | grep -E "us per"
python3 tools/probe_job.py 8192 4096 1920 1080 90 0:360:1 90 200 2>&1 | grep -E "us per"
