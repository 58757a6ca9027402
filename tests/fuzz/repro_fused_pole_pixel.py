#!/usr/bin/env python3
"""The one pixel per view where device-evaluated maps and NumPy's may disagree by half a panorama: the pole itself.
fuzz_fused.py case 125 of seed 74 (round 4): 1024 x 512 -> 372 x 183, FOV 90, pitch 9 -- output pixel (186, 62) has
x = u - W/2 = 0 exactly and looks 0.06 rows past the pole row; its azimuth is atan2(0, z') with z' a rounding residue of
the pitch rotation, whose SIGN decides between column 256 and column 768.  NumPy's sgemm (P:155) and the device's
multiply-adds round that residue differently; no other pixel of the view differs by more than 1.  Prints the differing
bytes and the NumPy map there.   python tests/fuzz/repro_fused_pole_pixel.py   (GPU box)"""
import importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import oracle_views, oracle_maps
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
pw, ph, ow, oh, fov = 1024, 512, 372, 183, 90
yaws, pitches = [285, 252], [9, 54]
pano = synth.synth_pano(pw, ph, 700 + 125, "S")
got = nat.remap_views(pano, yaws, pitches, fov, ow, oh)
want = oracle_views(pano, yaws, pitches, ow, oh, fov)
d = np.abs(got.astype(np.int16) - want.astype(np.int16))
idx = np.argwhere(d > 1)
print("bytes above 1:", idx.tolist())
rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
for (yi, pi, y, x, c) in idx[::3]:
    print("yaw", yaws[yi], "pitch", pitches[pi], "pixel", (int(x), int(y)), "pitch-map U,V =", float(U[pi][y, x]), float(V[pi][y, x]),
          "neighbours V:", [float(V[pi][yy, xx]) for yy in (y-1, y, y+1) for xx in (x-1, x, x+1) if 0 <= yy < oh and 0 <= xx < ow])
