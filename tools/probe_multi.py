#!/usr/bin/env python3
"""Diagnostic: config-4 geometry, one pitch per job -- one job run repeatedly vs two / three identical jobs run in
turn (what a multi-pitch launch does to the caches without being one)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
pano = synth.synth_pano(16384, 8192, 1000, "S")
ctx = nat.Context(0)
Y = list(range(0, 360, 5))
for njobs in (1, 2, 3):
    jobs = [nat.Job(ctx, 16384, 8192, 1, Y, [90], 60, 4096, 4096) for _ in range(njobs)]
    for j in jobs:
        j.set_pano(0, pano)
        j.time_launches(False)
    for _ in range(20):
        for j in jobs:
            j.run()
    ctx.mark(0)
    n = 20
    for _ in range(n):
        for j in jobs:
            j.run()
    ctx.mark(1)
    print("%d single-pitch jobs in turn: %.1f us per job launch" % (njobs, ctx.marked_ms() / n / njobs * 1e3), flush=True)
    for j in jobs:
        j.close()
