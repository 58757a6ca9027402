#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for cap in 12 18 24 36 48; do echo "max ppb $cap"; P2P_MAX_PAIRS_PER_BLOCK=$cap python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 30,60,90,120,150 8 2>&1 | grep -E "us per"; done
