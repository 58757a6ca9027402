for ow in 700 800 900 1024 1152 1280 1400; do
for v in "P2P_BAND=0" "P2P_BAND=1 P2P_TILE_SHAPE=64" "P2P_BAND=1" "P2P_BAND=1 P2P_BAND_BH=16 P2P_BAND_CW=8"; do
echo -n "$ow [$v]: "; env $v timeout 300 python3 tools/probe_job.py 8192 4096 $ow $ow 90 0,90,180,270 30,60,90,120,150 300 | tail -1 | cut -c50-90
done; done
