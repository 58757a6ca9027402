#!/usr/bin/env python3
"""A few launches of one job for profiler passes: python3 tools/run_job.py <n_yaw> <ppb> [pitches...]"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ny, ppb = int(sys.argv[1]), int(sys.argv[2])
pitches = [int(x) for x in sys.argv[3:]] or [60, 90, 120]
if ppb > 0:
    os.environ["P2P_PAIRS_PER_BLOCK"] = str(ppb)
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
pano = synth.synth_pano(8192, 4096, 1000, "S")
ctx = nat.Context(0)
job = nat.Job(ctx, 8192, 4096, 1, [(i * 360) // ny for i in range(ny)], pitches, 90, 1920, 1080)
job.set_pano(0, pano)
job.time_launches(8)
for _ in range(6):
    job.run()
ctx.synchronize()
print("kernel ms", job.kernel_ms())
