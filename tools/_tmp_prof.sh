cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1; do
rm -rf $R/gpurun_out/prof_cli_band$m
P2P_BAND_MERGE=$m rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cli_band$m -o cli --output-format csv -- python3 $R/tools/probe_job.py 8192 4096 800 800 90 0,90,180,270 30,60,90,120,150 300 > $R/gpurun_out/prof_cli_band$m.log 2>&1
tail -1 $R/gpurun_out/prof_cli_band$m.log
python3 - <<PY
import csv,glob,os
for f in glob.glob('$R/gpurun_out/prof_cli_band$m/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'remap_views' in r['Name']: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
# no-pole variant: pitches 60 90 120 only
P2P_BAND=1 python3 $R/tools/probe_job.py 8192 4096 800 800 90 0,90,180,270 60,90,120 300
P2P_BAND=0 python3 $R/tools/probe_job.py 8192 4096 800 800 90 0,90,180,270 60,90,120 300
P2P_BAND=1 python3 $R/tools/probe_job.py 8192 4096 800 800 90 0,90,180,270 30,150 300
P2P_BAND=0 python3 $R/tools/probe_job.py 8192 4096 800 800 90 0,90,180,270 30,150 300
