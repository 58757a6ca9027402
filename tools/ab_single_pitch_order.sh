for y in "0,30,60,90,120" "0:360:30"; do for mo in -1 1; do for t in 0 -1; do
echo -n "yaws $y main_order $mo tail $t: "; P2P_MAIN_ORDER=$mo P2P_MAIN_TAIL=$t timeout 120 python3 tools/probe_job.py 8192 4096 1920 1080 90 $y 60 800 2>&1 | tail -1 | sed 's/.*pitches: *//'
done; done; done
