#!/bin/bash
# the gather kernel's pairs per workgroup on the reference CLI's default view set (4 yaws x 5 pitches of an 8K panorama):
# with ONE yaw per workgroup the grid runs every tile of the XCD lists for yaw 0, then for yaw 1 ... (z slowest)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do for p in 16 2 1; do
  P2P_GATHER_PPB=$p timeout 200 python3 tools/probe_job.py 8192 4096 800 800 90 0,90,180,270 30,60,90,120,150 600 2>&1 | tail -1 | sed "s/^/ppb $p: /"
done; done
