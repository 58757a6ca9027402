#!/bin/bash
# pairs per workgroup (P2P_PAIRS_PER_BLOCK; 0 = the library's rule) on jobs with few tiles: the shares of the view-sharded
# path (4 / 5 / 9 yaws of one 1080p pitch view: 2040 tiles) and smaller views (1280x720: 900 tiles; 640x360: 230)
probe() { echo -n "$1 ppb $2: "; P2P_PAIRS_PER_BLOCK=$2 timeout 120 python3 tools/probe_job.py 8192 4096 $3 $4 $5 $6 $7 600 2>&1 | tail -1 | sed 's/.*pitches: *//'; }
for y in "0,30,60,90,120" "0,30,60,90" "0,30,60,90,120,150,180,210,240"; do for ppb in 0 1 2 3 5 9; do probe "1920x1080 yaws $y" $ppb 1920 1080 90 $y 60; done; done
for ppb in 0 2 3 4 6 12; do probe "1280x720 fov 60, 12 yaws x 1 pitch" $ppb 1280 720 60 0:360:30 90; done
for ppb in 0 1 2 3 4 6 12; do probe "640x360 fov 30, 12 yaws x 1 pitch" $ppb 640 360 30 0:360:30 90; done
for ppb in 0 2 3 4 6 12; do probe "1280x720 fov 60, 12 yaws x 3 pitches" $ppb 1280 720 60 0:360:30 60,90,120; done
