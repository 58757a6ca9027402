#!/usr/bin/env python3
"""Diagnostic: per-launch time of one job geometry.
    python tools/probe_job.py PW PH OW OH FOV YAWS PITCHES [launches]      YAWS / PITCHES: comma lists or a:b:step
Under `rocprofv3 --kernel-trace --stats` the split over the view kernels shows."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")


def lst(s):
    if ":" in s:
        a, b, c = (int(v) for v in s.split(":"))
        return list(range(a, b, c))
    return [float(v) if "." in v else int(v) for v in s.split(",")]


pw, ph, ow, oh, fov = (int(v) for v in sys.argv[1:6])
yaws, pitches = lst(sys.argv[6]), lst(sys.argv[7])
n = int(sys.argv[8]) if len(sys.argv) > 8 else 300
pano = synth.synth_pano(pw, ph, 1000, "S")
ctx = nat.Context(0)
job = nat.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
job.set_pano(0, pano)
job.time_launches(False)
for _ in range(max(20, n // 3)):
    job.run()
ctx.mark(0)
for _ in range(n):
    job.run()
ctx.mark(1)
ms = ctx.marked_ms() / n
npx = len(yaws) * len(pitches) * ow * oh
b_alg = 3 * pw * ph + 3 * npx
print("%dx%d -> %dx%d fov %d, %d yaws x %d pitches: %8.1f us per launch  %7.1f Gpix/s  %.3f of 8 TB/s" %
      (pw, ph, ow, oh, fov, len(yaws), len(pitches), ms * 1e3, npx / ms / 1e6, b_alg / ms / 1e6 / 8000.0), flush=True)
job.close(); ctx.close()
