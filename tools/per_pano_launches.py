#!/usr/bin/env python3
"""N panoramas x 36 views (config 3's share of a GPU): ONE job of N resident panoramas (one launch) against N jobs of one
panorama each, launched one after the other on one stream, or alternating on two streams (two contexts).
    python tools/per_pano_launches.py [--panos 8] [--rounds 40]"""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
ap = argparse.ArgumentParser(description=__doc__, allow_abbrev=False)
ap.add_argument("--panos", type=int, default=8)
ap.add_argument("--rounds", type=int, default=40)
a = ap.parse_args()
pw, ph, ow, oh, fov = 8192, 4096, 1920, 1080, 90
yaws, pitches = list(range(0, 360, 30)), [60, 90, 120]
panos = [synth.synth_pano(pw, ph, 1000 + i, "S") for i in range(a.panos)]
ctxs = [nat.Context(0) for _ in range(2)]


def sync():
    for c in ctxs:
        c.synchronize()


def timed(fn):
    for _ in range(5):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.rounds):
        fn()
    sync()
    return (time.perf_counter() - t0) / a.rounds * 1e6


big = nat.Job(ctxs[0], pw, ph, a.panos, yaws, pitches, fov, ow, oh)
for i, p in enumerate(panos):
    big.set_pano(i, p)
t_big = timed(big.run)
print("one job, %d resident panoramas, one launch:            %8.1f us (%.1f per panorama), tile width %d" %
      (a.panos, t_big, t_big / a.panos, big.info()["tile_w"]), flush=True)
big.close()
for nctx in (1, 2):
    jobs = []
    for i, p in enumerate(panos):
        j = nat.Job(ctxs[i % nctx], pw, ph, 1, yaws, pitches, fov, ow, oh)
        j.set_pano(0, p)
        jobs.append(j)
    t = timed(lambda: [j.run() for j in jobs])
    print("%d jobs of one panorama, %d stream(s):                   %8.1f us (%.1f per panorama)" % (a.panos, nctx, t, t / a.panos), flush=True)
    for j in jobs:
        j.close()
for c in ctxs:
    c.close()
