#!/usr/bin/env python3
"""Randomised look at the fused path's tolerance: in-kernel maps vs the NumPy-map oracle on band-limited ("S")
panoramas.  Prints, per case, the largest channel difference and the fraction of differing bytes; north_star's
bar is +-1.  Usage: python tests/fuzz/fuzz_fused.py [n_cases] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import oracle_maps, oracle_views
from test_gpu_fused_exceptions import pole_pixels, seam_pixels  # the masks of the mode's pinned exceptions
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
import _args  # named options with hard caps (tests/fuzz/_args.py)
_a = _args.parser(__doc__, cases=100, seed=3).parse_args()
n_cases, seed = _a.cases, _a.seed
worst = 0; over = 0; t0 = time.time()
for case in range(int(os.environ.get("FUZZ_FIRST", "0")), n_cases):  # FUZZ_FIRST: resume a long run
    rng = np.random.default_rng(seed * 100003 + case)
    pw = int(rng.choice([512, 1024, 2048, 4096])); ph = pw // 2
    ow, oh = int(rng.integers(16, 500)), int(rng.integers(16, 400))
    fov = int(rng.choice([20, 45, 60, 90, 90, 120, 150, 170]))
    yaws = [int(v) for v in rng.integers(-30, 400, size=int(rng.integers(1, 4)))]
    pitches = [int(v) for v in rng.integers(1, 180, size=int(rng.integers(1, 3)))]
    pano = synth.synth_pano(pw, ph, 700 + case, "S")
    got = nat.remap_views(pano, yaws, pitches, fov, ow, oh)
    want = oracle_views(pano, yaws, pitches, ow, oh, fov)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    m = int(d.max()); frac = float((d > 0).mean()); big = int((d > 1).sum())
    worst = max(worst, m)
    if m > 1:
        over += 1
        # where: the pole pixel, the seam row, or elsewhere (tests/test_gpu_fused_exceptions.py has the definitions)
        _, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
        px = d.max(axis=-1) > 1  # [yaw][pitch][oh][ow]
        n_pole = n_seam = n_else = 0
        for pi in range(len(pitches)):
            pole = pole_pixels(U[pi], V[pi], ow, ph)
            seam = seam_pixels(U[pi], pw)
            seam = seam & (seam.sum(axis=1, keepdims=True) >= ow // 4)
            for yi in range(len(yaws)):
                n_pole += int((px[yi, pi] & pole).sum()); n_seam += int((px[yi, pi] & seam & ~pole).sum())
                n_else += int((px[yi, pi] & ~pole & ~seam).sum())
        print("case %d: max diff %d (%d bytes > 1 of %d; pixels: %d pole, %d seam row, %d elsewhere), differing %.4f  "
              % (case, m, big, d.size, n_pole, n_seam, n_else, frac),
              dict(pw=pw, ow=ow, oh=oh, fov=fov, yaws=yaws, pitches=pitches), flush=True)
print("fuzz_fused finished: %d cases, worst difference %d, %d cases above 1, %.0f s" % (n_cases, worst, over, time.time() - t0))
