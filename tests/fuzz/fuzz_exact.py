#!/usr/bin/env python3
"""Randomised check of the IDENTICAL-RESULTS mode: process_views(..., exact=True) -- pitch maps evaluated on the host by the
package's NumPy builder, yaw tables and every pixel on the GPU -- against the oracle on NOISE panoramas, where a coordinate
that falls on the other side of a 1/32-pixel tie would show.  Geometries as fuzz_fused.py draws them (poles in view, seam
rows, FOV 20..170, real-valued and out-of-range yaws); the module's caches and the one-shot slots' named maps carry over from
case to case, as in a long-lived process.  Bar: 0 differing bytes.
    python tests/fuzz/fuzz_exact.py --cases 300 --seed 5"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import oracle_views
pkg = importlib.import_module("360-to-planer-images_amd"); tool = pkg.panorama_to_plane_pitch
synth = importlib.import_module("360-to-planer-images_amd.synth")
import _args  # named options with hard caps (tests/fuzz/_args.py)
_a = _args.parser(__doc__, cases=100, seed=5).parse_args()
n_cases, seed = _a.cases, _a.seed
bad = 0; t0 = time.time()
panos = {}
for case in range(n_cases):
    rng = np.random.default_rng(seed * 100003 + case)
    pw = int(rng.choice([512, 1024, 2048, 4096])); ph = pw // 2
    ow, oh = int(rng.integers(16, 500)), int(rng.integers(16, 400))
    fov = int(rng.choice([20, 45, 60, 90, 90, 120, 150, 170]))
    yaws = [float(np.round(v, 1)) if case % 2 else int(v) for v in rng.uniform(-30, 400, size=int(rng.integers(1, 4)))]
    pitches = sorted(set(int(v) for v in rng.integers(1, 180, size=int(rng.integers(1, 4)))))
    if case % 7 == 0:  # the coincidences the fused mode's exceptions come from: a seam row, a pole pixel
        fov, pitches, ow = 90, [45] + pitches[:1], ow | 1
        oh = oh | 1
    pano = panos.setdefault(pw, synth.synth_pano(pw, ph, 900 + pw, "N"))
    got = tool.process_views(pano, yaws, pitches, ow, oh, fov, exact=True)
    want = oracle_views(pano, yaws, pitches, ow, oh, fov)
    n = int((got != want).sum())
    if n:
        bad += 1
        print("case %d: %d differing bytes of %d  " % (case, n, want.size), dict(pw=pw, ow=ow, oh=oh, fov=fov, yaws=yaws, pitches=pitches), flush=True)
print("fuzz_exact finished: %d cases, %d with differing bytes, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
