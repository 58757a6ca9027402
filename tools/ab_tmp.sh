cd $GRAFT_REPO_ROOT
P5="python3 tools/probe_job.py 16384 8192 4096 4096 60 0:360:5"
for g in 48 96 144 192 384; do echo "cfg4 w128 group $g"; P2P_MAIN_GROUP=$g timeout 200 $P5 30,60,90,120,150 6 2>&1 | grep "us per"; done
echo "cfg4 w128 no prefetch"; P2P_PREFETCH_LEAD=0 timeout 200 $P5 30,60,90,120,150 6 2>&1 | grep "us per"
echo "cfg4 w128 grid order"; P2P_MAIN_ORDER=0 timeout 200 $P5 30,60,90,120,150 6 2>&1 | grep "us per"
