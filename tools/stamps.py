#!/usr/bin/env python3
"""Diagnostic (library built with P2P_EXTRA_FLAGS=-DP2P_STAMPS): where a wave of remap_views_kernel
spends its cycles per (panorama, yaw) pair.  Never quote this build's run time, only the shares."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module(bench.PKG); nat = pkg._native
synth = importlib.import_module(bench.PKG + ".synth")
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
ctx = nat.Context(0)
job = nat.Job(ctx, w["pw"], w["ph"], 1, w["yaws"], w["pitches"], w["fov"], w["ow"], w["oh"])
job.set_pano(0, synth.synth_pano(w["pw"], w["ph"], 1000, "S"))
for _ in range(3):
    job.run()
ctx.synchronize()
nat.debug_stamps(True)
job.run(); ctx.synchronize()
st = nat.debug_stamps(True)
names = ["load-wait + stage 1 + LDS write", "barrier", "next pair ctx + issue loads", "LDS tap reads (wait)", "stage 2 blend", "store"]
waves, iters = int(st[6]), int(st[7])
print("waves", waves, "pair iterations (per wave)", iters, "kernel ms", job.kernel_ms())
tot = sum(int(x) for x in st[:6])
for n, x in zip(names, st[:6]):
    print("%-36s %8.0f cycles per wave-pair  %5.1f %%" % (n, int(x) / max(iters, 1), 100.0 * int(x) / max(tot, 1)))
print("%-36s %8.0f" % ("total", tot / max(iters, 1)))
