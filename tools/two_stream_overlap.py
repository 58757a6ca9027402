#!/usr/bin/env python3
"""Do two resident jobs on two contexts (two HIP streams) overlap at the ends of their launches?  About 8 us of a launch of config 2
do not scale with its tiles: its start and its drain.  Launches of ONE stream run one after the other;
launches of two streams may run side by side.
    python tools/two_stream_overlap.py [--launches 2000]"""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
ap = argparse.ArgumentParser(description=__doc__, allow_abbrev=False)
ap.add_argument("--launches", type=int, default=2000)
a = ap.parse_args()
pw, ph, ow, oh, fov = 8192, 4096, 1920, 1080, 90
yaws, pitches = list(range(0, 360, 30)), [60, 90, 120]
ctxs = [nat.Context(0) for _ in range(2)]
jobs = []
for i, c in enumerate(ctxs):
    j = nat.Job(c, pw, ph, 1, yaws, pitches, fov, ow, oh)
    j.set_pano(0, synth.synth_pano(pw, ph, 1000 + i, "S"))
    jobs.append(j)


def timed(which, n):
    for j in which:
        for _ in range(50):
            j.run()
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // len(which)):
        for j in which:
            j.run()
    for c in ctxs:
        c.synchronize()
    return (time.perf_counter() - t0) / (n // len(which) * len(which)) * 1e6


for rep in range(3):
    one = timed(jobs[:1], a.launches)
    two = timed(jobs, a.launches)
    print("one stream %.1f us per launch; two jobs on two streams, alternating: %.1f us per launch (%.1f %%)" %
          (one, two, 100.0 * (two - one) / one), flush=True)
for j in jobs:
    j.close()
for c in ctxs:
    c.close()
