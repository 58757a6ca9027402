R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in dw4 dw3 dw4 dw3; do
rm -rf $R/gpurun_out/legacy_$v
P2P_LIB_PATH=$R/gpurun_variants/libp2p_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/legacy_$v -- python3 $R/tools/legacy_remap_time.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob('$R/gpurun_out/legacy_$v/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'table_kernel' in r['Name'] or 'remap_maps' in r['Name']: print('$v', r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
