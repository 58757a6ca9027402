#!/usr/bin/env python3
"""Diagnostic: device time of the plan pass and of the yaw tables for the BASELINE geometries (fresh tables every time:
P2P_PLAN_CACHE=0), median of a few builds.      python tools/plan_times.py [builds]"""
import importlib, os, sys
os.environ["P2P_PLAN_CACHE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
GEOS = {"cfg2": (8192, 4096, 1920, 1080, 90, list(range(0, 360, 30)), [60, 90, 120]),
        "cfg4": (16384, 8192, 4096, 4096, 60, list(range(0, 360, 5)), [30, 60, 90, 120, 150]),
        "cfg5": (8192, 4096, 1920, 1080, 90, list(range(360)), [90]),
        "cli": (8192, 4096, 800, 800, 90, [0, 90, 180, 270], [30, 60, 90, 120, 150]),
        "cfg1": (2048, 1024, 512, 512, 90, [0], [90])}
ctx = nat.Context(0)
for name, (pw, ph, ow, oh, fov, yaws, pitches) in GEOS.items():
    pano = np.zeros((ph, pw, 3), np.uint8)
    pl, tb = [], []
    for _ in range(n):
        job = nat.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
        job.set_pano(0, pano)
        job.time_launches(1)  # (the plan pass is timed for jobs that ask)
        job.run()
        a, b = job.plan_ms()
        pl.append(a); tb.append(b)
        job.close()
    print("%-5s plan pass %8.1f us (min %.1f)   yaw tables %7.1f us   [%d px x %d pitch views]" %
          (name, 1e3 * float(np.median(pl)), 1e3 * min(pl), 1e3 * float(np.median(tb)), ow * oh, len(pitches)), flush=True)
ctx.close()
