#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for side in 0 1; do echo "side stream $side"; P2P_SIDE_STREAM=$side python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 30 20 2>&1 | grep -E "us per"; done
for side in 0 1; do echo "side stream $side"; P2P_SIDE_STREAM=$side python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 30,60,90,120,150 10 2>&1 | grep -E "us per"; done
for cap in 24 36 64; do echo "max ppb $cap"; P2P_MAX_PAIRS_PER_BLOCK=$cap python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 30,60,90,120,150 10 2>&1 | grep -E "us per"; done
