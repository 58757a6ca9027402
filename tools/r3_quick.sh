#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_views_exact.py tests/test_gpu_views_fused.py -m gpu -q -x --timeout 600 2>&1 | tail -4
python3 tools/probe_default_cli.py 2>&1 | tail -5
