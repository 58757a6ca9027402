#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export P2P_VERBOSE=1
for p in 30 60 90; do python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 $p 20 2>&1 | grep -E "plan|us per"; done
python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 30,60,90,120,150 10 2>&1 | grep -E "plan|us per"
