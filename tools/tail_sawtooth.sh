#!/bin/bash
# does the last, partly filled round of workgroups cost a whole round?  config 2 with the view height varied: tiles =
# 30 x ceil(oh / 16) x 3 pitch views; a GPU holds 1792 workgroups of the 64-wide kernel at a time
for oh in 896 944 960 976 1024 1072 1088 1136 1184 1232 1264 1280 1296; do
  t=$(timeout 120 python3 tools/probe_job.py 8192 4096 1920 $oh 90 0:360:30 60,90,120 600 2>&1 | tail -1 | sed 's/.*pitches: *\([0-9.]*\) us.*/\1/')
  python3 -c "
import sys,math
oh=int(sys.argv[1]); t=float(sys.argv[2]); tiles=30*math.ceil(oh/16)*3
print('oh %4d  tiles %5d = %.2f rounds of 1792   %7.1f us per launch   %.2f ns per tile   %.2f us per round-equivalent' % (oh, tiles, tiles/1792, t, 1e3*t/tiles, t/(tiles/1792)))" $oh $t
done
