#!/usr/bin/env python3
"""Diagnostic: the reference CLI's DEFAULT view set (P:412-437: 800 x 800, FOV 90, yaw 0/90/180/270, pitch
30/60/90/120/150 -- the pitch 30 / 150 views hold a pole) on an 8K panorama; per-launch time by pitch subset.
Under rocprofv3 --kernel-trace the main / rest kernel split shows."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
pw, ph = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 4096)
pano = synth.synth_pano(pw, ph, 1000, "S")
ctx = nat.Context(0)
for pitches in ([30, 60, 90, 120, 150], [60, 90, 120], [30, 150], [30], [90]):
    job = nat.Job(ctx, pw, ph, 1, [0, 90, 180, 270], pitches, 90, 800, 800)
    job.set_pano(0, pano)
    job.time_launches(False)
    for _ in range(200):
        job.run()
    ctx.mark(0)
    n = 500
    for _ in range(n):
        job.run()
    ctx.mark(1)
    ms = ctx.marked_ms() / n
    npx = 4 * len(pitches) * 800 * 800
    print("pitches %-24s %7.1f us per launch  %7.1f Gpix/s" % (pitches, ms * 1e3, npx / ms / 1e6), flush=True)
    job.close()
