#!/usr/bin/env python3
"""Fuzz of the opt-in float pixel path: random geometries (incl. pole views, odd sizes, pixel centres, real-valued
angles); float16 vs float32 within 1 level everywhere, float32 vs the NumPy float32 evaluation of the same formula
within 1 level (2 allowed next to a pole, where the device map and NumPy's differ by more than an ulp), and the
view-sharded driver against the single-device result.  Usage: python tests/fuzz/fuzz_float.py [n_cases] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_float_path import numpy_float_views
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
drv = importlib.import_module("360-to-planer-images_amd._driver")
synth = importlib.import_module("360-to-planer-images_amd.synth")
import _args  # named options with hard caps (tests/fuzz/_args.py)
_a = _args.parser(__doc__, cases=100, seed=77).parse_args()
n_cases, seed = _a.cases, _a.seed
t0 = time.time(); bad = 0; worst16 = 0; worst32 = 0
for case in range(int(os.environ.get("FUZZ_FIRST", "0")), n_cases):  # FUZZ_FIRST: resume a long run
    rng = np.random.default_rng(seed * 7919 + case)
    pw = int(rng.choice([256, 512, 1000, 1024, 2048, 4096])); ph = max(16, pw // 2)
    ow, oh = int(rng.integers(8, 500)), int(rng.integers(8, 300))
    fov = float(rng.choice([30, 60, 90, 120, 150])) if rng.random() < 0.7 else float(rng.uniform(20, 150))
    yaws = [float(v) for v in rng.uniform(-360, 720, size=int(rng.integers(1, 20)))]
    pitches = [float(v) for v in rng.uniform(1, 179, size=int(rng.integers(1, 4)))]
    kind = "S" if rng.random() < 0.5 else "N"
    pano = synth.synth_pano(pw, ph, 7000 + case, kind)
    f32 = nat.remap_views_f64(pano, yaws, pitches, fov, ow, oh, flags=nat.FLAG_PIXELS_F32)
    f16 = nat.remap_views_f64(pano, yaws, pitches, fov, ow, oh, flags=nat.FLAG_PIXELS_F16)
    d16 = int(np.abs(f32.astype(int) - f16.astype(int)).max()); worst16 = max(worst16, d16)
    msg = []
    if d16 > 1:
        msg.append("f16 vs f32 %d" % d16)
    if kind == "S":
        want = numpy_float_views(pano, yaws, pitches, ow, oh, fov)
        d = np.abs(f32.astype(int) - want.astype(int))
        d32 = int(d.max()); worst32 = max(worst32, d32)
        if d32 > 2 or (d > 1).mean() > 1e-3:
            msg.append("f32 vs numpy max %d, >1: %.2g" % (d32, (d > 1).mean()))
    if case % 5 == 0:
        sh = drv.process_views_sharded(pano, yaws, pitches, ow, oh, fov, [0, 0, 0], flags=nat.FLAG_PIXELS_F16)
        if not np.array_equal(sh, f16):
            msg.append("sharded != single")
    if msg:
        bad += 1
        print("MISMATCH", dict(case=case, pw=pw, ow=ow, oh=oh, fov=fov, yaws=yaws[:3], pitches=pitches, kind=kind), msg, flush=True)
    if case % 20 == 19:
        print("case %d done, %.0f s, bad %d, worst f16-f32 %d, worst f32-numpy %d" % (case + 1, time.time() - t0, bad, worst16, worst32), flush=True)
print("fuzz_float finished: %d cases, %d bad, worst f16-f32 %d, worst f32-numpy %d, %.0f s" % (n_cases, bad, worst16, worst32, time.time() - t0))
sys.exit(1 if bad else 0)
