#!/usr/bin/env python3
"""Parity numbers for DESIGN.md (run on the GPU box): coordinate flips of the in-kernel maps against the
oracle's NumPy maps, and pixel differences of the fused path on band-limited and noise panoramas."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import diff_stats, oracle_views
from oracle import cpu_ref, maps
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
rep = {"maps": [], "pixels": []}
for (ow, oh, pitch, pw, ph, fov) in [(512, 512, 90, 2048, 1024, 90), (1920, 1080, 60, 8192, 4096, 90),
                                     (1920, 1080, 90, 8192, 4096, 90), (1920, 1080, 120, 8192, 4096, 90),
                                     (800, 800, 30, 4096, 2048, 90), (4096, 4096, 60, 16384, 8192, 60)]:
    U, V = nat.build_pitch_map(ow, oh, np.radians(fov), np.radians(pitch), pw, ph)
    Ur, Vr = maps.pitch_map_deg(ow, oh, pitch, pw, ph, fov)
    ok = ~(np.isnan(V) | np.isnan(Vr))
    sx, sy, _, _ = cpu_ref.quantise_maps(U, V); rx, ry, _, _ = cpu_ref.quantise_maps(Ur, Vr)
    dU = np.abs(U - Ur); dU = np.minimum(dU, pw - 1 - dU)
    rep["maps"].append({"cfg": [ow, oh, pitch, pw, ph, fov],
                        "bit_equal_U": float((U == Ur)[ok].mean()), "bit_equal_V": float((V == Vr)[ok].mean()),
                        "max_dU_px": float(dU[ok].max()), "max_dV_px": float(np.abs(V - Vr)[ok].max()),
                        "coord_flips": float(((sx != rx) | (sy != ry))[ok].mean())})
for (pw, ph, ow, oh, fov, yaws, pitches) in [(2048, 1024, 512, 512, 90, [0], [90]),
                                             (8192, 4096, 1920, 1080, 90, [0, 30], [60, 90, 120])]:
    for kind in ("S", "N"):
        pano = synth.synth_pano(pw, ph, 1000, kind)
        got = pkg.process_views(pano, yaws, pitches, ow, oh, fov)
        want = oracle_views(pano, yaws, pitches, ow, oh, fov)
        mx, gt1, anyd = diff_stats(got, want)
        rep["pixels"].append({"cfg": [pw, ph, ow, oh, fov, yaws, pitches], "kind": kind, "max_abs_diff": mx,
                              "frac_gt1": gt1, "frac_any": anyd})
print(json.dumps(rep, indent=1))
