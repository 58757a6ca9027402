#!/bin/bash
# what of a launch does not scale with its tiles: config 2 with its three pitch views listed 1 .. 4 times (the same mix of
# tiles; from 3 x on the views no longer fit the Infinity Cache), with and without the tail split of the lists
for p in "60,90,120" "60,90,120,60,90,120" "60,90,120,60,90,120,60,90,120" "60,90,120,60,90,120,60,90,120,60,90,120"; do
echo -n "pitches $p: "; timeout 120 python3 tools/probe_job.py 8192 4096 1920 1080 90 0:360:30 $p 600 2>&1 | tail -1 | sed 's/.*pitches: *//'
done
for p in "60,90,120" "60,90,120,60,90,120"; do
echo -n "tail 0, pitches $p: "; P2P_MAIN_TAIL=0 timeout 120 python3 tools/probe_job.py 8192 4096 1920 1080 90 0:360:30 $p 600 2>&1 | tail -1 | sed 's/.*pitches: *//'
done
