#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
echo "five times pitch 90 in one job"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 90,90,90,90,90 8 2>&1 | grep -E "us per"
echo "pitch 90 alone"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 90 20 2>&1 | grep -E "us per"
echo "pitch 60,90,120"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5 60,90,120 10 2>&1 | grep -E "us per"
echo "pitch 90, 36 yaws"; python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:10 90 20 2>&1 | grep -E "us per"
echo "pitch 90, 8K pano (fits the Infinity Cache), 72 yaws"; python3 tools/probe_job.py 8192 4096 4096 4096 60 0:360:5 90 20 2>&1 | grep -E "us per"
