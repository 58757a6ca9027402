for y in "0,30,60,90,120" "0,30,60,90" "0,30,60,90,120,150,180,210,240"; do for ppb in 0 1 2 3 5 9; do
echo -n "yaws $y ppb $ppb: "; P2P_PAIRS_PER_BLOCK=$ppb timeout 120 python3 tools/probe_job.py 8192 4096 1920 1080 90 $y 60 600 2>&1 | tail -1
done; done
