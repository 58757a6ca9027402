#!/usr/bin/env python3
"""One-off fuzz of p2p_remap_maps_interp_u8 (generic cv2.remap: three interpolations x five border modes x
1 / 3 / 4 channels, random sizes and maps incl. NaN / huge / out-of-range coordinates) and of the fused view path's
self-consistency (in-kernel coordinates re-fed as caller maps reproduce the fused output).
Usage: python tests/fuzz/fuzz_remap.py [n_cases] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import coords_to_maps
from oracle import cpu_ref, maps
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
import _args  # named options with hard caps (tests/fuzz/_args.py)
_a = _args.parser(__doc__, cases=100, seed=5).parse_args()
n_cases, seed = _a.cases, _a.seed
part = os.environ.get("FUZZ_REMAP_PART", "both")  # remap | job | both: which half runs (hunting a rare fault)
bad = 0; t0 = time.time()
for case in range(int(os.environ.get("FUZZ_FIRST", "0")), n_cases):  # FUZZ_FIRST: resume a long run
    rng = np.random.default_rng(seed * 100003 + case)
    cn = int(rng.choice([1, 3, 3, 4])); interp = int(rng.choice([0, 1, 1, 2])); mode = int(rng.integers(0, 5))
    sh, sw = int(rng.integers(1, 300)), int(rng.integers(1, 400))
    oh, ow = int(rng.integers(1, 200)), int(rng.integers(1, 260))
    img = rng.integers(0, 256, size=(sh, sw, cn), dtype=np.uint8)
    kind = rng.random()
    if kind < 0.4:      # wild
        U = rng.uniform(-2 * sw - 5, 3 * sw + 5, size=(oh, ow)).astype(np.float32)
        V = rng.uniform(-2 * sh - 5, 3 * sh + 5, size=(oh, ow)).astype(np.float32)
    elif kind < 0.8:    # smooth, mostly inside
        yy, xx = np.mgrid[0:oh, 0:ow].astype(np.float32)
        U = (rng.uniform(-3, 3) + xx * rng.uniform(0.2, 2.0) * sw / max(ow, 1) + yy * rng.uniform(-0.3, 0.3)).astype(np.float32)
        V = (rng.uniform(-3, 3) + yy * rng.uniform(0.2, 2.0) * sh / max(oh, 1) + xx * rng.uniform(-0.3, 0.3)).astype(np.float32)
    else:               # exact grid points and ties
        U = rng.integers(-2, sw + 2, size=(oh, ow)).astype(np.float32) + rng.choice([0.0, 0.5, 0.015625], size=(oh, ow)).astype(np.float32)
        V = rng.integers(-2, sh + 2, size=(oh, ow)).astype(np.float32) + rng.choice([0.0, 0.5, 0.984375], size=(oh, ow)).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):
        U[rng.integers(0, oh), rng.integers(0, ow)] = rng.choice([np.nan, np.inf, -np.inf, 1e9, -1e9, 40000.0])
        V[rng.integers(0, oh), rng.integers(0, ow)] = rng.choice([np.nan, np.inf, -np.inf, 1e9, -1e9, -40000.0])
    cval = rng.integers(0, 256, size=4, dtype=np.uint8) if rng.random() < 0.7 else None
    if part != "job":
        got = nat.remap_maps(img, U, V, border=mode, border_value=cval, interpolation=interp)
        want = cpu_ref.remap(img, U, V, mode, cval, interpolation=interp)
    else:
        got = want = np.zeros(1)
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH remap", dict(case=case, cn=cn, interp=interp, mode=mode, sh=sh, sw=sw, oh=oh, ow=ow, kind=float(kind),
                                     n=int((got != want).sum())), flush=True)
    if case % 5 == 0 and part != "remap":   # fused self-consistency on a random small job
        pw = int(rng.choice([256, 512, 1024, 2048])); ph = pw // 2
        vw, vh = int(rng.integers(16, 300)), int(rng.integers(16, 200))
        yaws = [int(v) for v in rng.integers(0, 360, size=int(rng.integers(1, 14)))]
        pitches = [int(v) for v in rng.integers(5, 176, size=int(rng.integers(1, 4)))]
        fov = int(rng.choice([60, 90, 120]))
        n_panos = int(rng.integers(1, 3))
        ctx = nat.Context(0)
        job = nat.Job(ctx, pw, ph, n_panos, yaws, pitches, fov, vw, vh)
        panos = [synth.synth_pano(pw, ph, 100 + case + i, "N") for i in range(n_panos)]
        for i, p in enumerate(panos):
            job.set_pano(i, p)
        job.run()
        fused = [job.get_views(i) for i in range(n_panos)]
        coords = job.get_coords()
        job.close(); ctx.close()
        UV = [coords_to_maps(coords[p]) for p in range(len(pitches))]
        rows = np.stack([maps.yaw_column_table(pw, y) for y in yaws])
        for i, p in enumerate(panos):
            again = nat.remap_views_maps(p, rows, np.stack([u for u, _ in UV]), np.stack([v for _, v in UV]))
            if not np.array_equal(again, fused[i]):
                bad += 1
                print("MISMATCH fused-vs-refed", dict(case=case, pw=pw, vw=vw, vh=vh, yaws=yaws, pitches=pitches, fov=fov, pano=i,
                                                      n=int((again != fused[i]).sum())), flush=True)
print("fuzz_remap finished: %d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
