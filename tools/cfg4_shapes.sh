#!/bin/bash
# Config 4 launch shapes (one pitch view vs five in one job, output size vs pitch count, chunk lengths): what
# profiles/r03_cfg4_one_vs_five_pitches_counters.txt quotes.  GPU box, repo root: bash tools/cfg4_shapes.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
echo "five times pitch 90 in one job"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 90,90,90,90,90 8 2>&1 | grep -E "us per"
echo "pitch 90 alone"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 90 20 2>&1 | grep -E "us per"
echo "pitch 60,90,120"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 60,90,120 10 2>&1 | grep -E "us per"
echo "pitch 90, 36 yaws"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:10 90 20 2>&1 | grep -E "us per"
echo "pitch 90, 8K pano (fits the Infinity Cache), 72 yaws"; python3 tools/probe_job.py 8192 4096 4096 4096 60 0:360:5 90 20 2>&1 | grep -E "us per"
echo "pitch 90, 360 yaws (18 GB of views, one pitch)"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:1 90 6 2>&1 | grep -E "us per"
echo "five pitches, 14 yaws (3.5 GB of views)"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:350:25 30,60,90,120,150 10 2>&1 | grep -E "us per"
echo "five times pitch 90, 14 yaws"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:350:25 90,90,90,90,90 10 2>&1 | grep -E "us per"
echo "pitch 90, 72 yaws, chunks of 24"; P2P_MAX_PAIRS_PER_BLOCK=24 python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 90 20 2>&1 | grep -E "us per"
for cap in 12 18 24 36 48; do echo "max pairs per workgroup $cap"; P2P_MAX_PAIRS_PER_BLOCK=$cap python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 30,60,90,120,150 8 2>&1 | grep -E "us per"; done
