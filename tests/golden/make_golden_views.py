#!/usr/bin/env python3
"""Generate tests/golden/views_golden.npz by RUNNING THE REFERENCE with the real OpenCV.

    pip install opencv-python-headless==4.10.0.84 numpy tqdm      (the reference's pins: /root/reference/pyproject.toml)
    python tests/golden/make_golden_views.py [--reference /path/to/360-to-planer-images]

This is the one piece of parity evidence that cannot be produced in the build container: cv2 is not installable
there (no wheel, no network), so the reference's gather (cv2.remap at P:192-199, P:212-218, L:179) cannot run and
the gather oracle is PARITY UNPINNED.  On any machine with the reference checked out and opencv-python installed this
script imports the reference's own modules UNSTUBBED, calls

    process_yaw_and_pitchs(pano, yaw, pitches, ow, oh, fov)      app/panorama_to_plane-pitch.py:181-221
    panorama_to_plane(pano, U, V)                                app/legacy/panorama_to_plane.py:182-194
    interpolate_color(U, V, pano, method)                        app/legacy/panorama_to_plane.py:159-180

on small seeded panoramas (noise, band-limited, with pole / seam / NaN-pixel views) and stores inputs and outputs.
tests/test_golden_views.py then checks the oracle against the file (CPU) and the HIP kernels against it (GPU);
until the file exists those tests say so loudly.  The output is DATA (inputs + the reference's outputs): no
reference source is stored.  It refuses to run with a stubbed or missing cv2.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "views_golden.npz")

VIEW_CASES = [  # (name, pw, ph, seed, kind, yaws, pitches, ow, oh, fov)
    ("noise_small", 512, 256, 2000, "N", [0, 30, 77, 359, -30], [30, 60, 90, 120, 150], 96, 64, 90),
    ("smooth_small", 512, 256, 2001, "S", [0, 45, 200], [60, 90, 120], 128, 72, 90),
    ("poles_and_nan", 1024, 512, 2002, "N", [0, 123], [1, 5, 12, 168, 175, 179], 160, 90, 90),
    ("wide_narrow_fov", 512, 256, 2003, "N", [15], [45, 90], 80, 80, 150),
    ("narrow_fov", 512, 256, 2004, "N", [15], [45, 90], 80, 80, 20),
    ("cli_defaults", 2048, 1024, 2005, "N", [0, 90, 180, 270], [30, 60, 90, 120, 150], 200, 200, 90),
    ("odd_sizes", 333, 111, 2006, "N", [10, 200], [30, 90, 150], 65, 47, 120),
]
LEGACY_CASES = [  # (name, pw, ph, seed, yaw, pitch, fov, ow, oh)
    ("legacy_a", 512, 256, 2100, 0, 90, 90, 96, 64),
    ("legacy_b", 512, 256, 2101, 77, 60, 100, 96, 64),
    ("legacy_pole", 512, 256, 2102, 30, 175, 90, 96, 64),
]


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        sys.exit("cv2 is not importable: this generator needs the real OpenCV (opencv-python[-headless]==4.10.0.84)")
    if not hasattr(cv2, "remap") or not hasattr(cv2, "__version__"):
        sys.exit("the cv2 module in sys.modules is a stub: refusing to write goldens that the reference did not produce")
    out = OUT
    if getattr(cv2, "__p2p_fake__", False):
        # tests/test_golden_views.py dry-runs this script's plumbing with an oracle-backed stand-in for cv2;
        # such a run may never produce the real fixture
        out = os.environ.get("P2P_GOLDEN_DRYRUN_OUT")
        if not out or os.path.abspath(out) == os.path.abspath(OUT):
            sys.exit("a stand-in cv2 is loaded: refusing to write tests/golden/views_golden.npz")
    sys.path.insert(0, ROOT)
    synth = importlib.import_module("360-to-planer-images_amd.synth")  # seeded synthetic panoramas (no cv2, no GPU)
    P = load(os.path.join(args.reference, "app", "panorama_to_plane-pitch.py"), "ref_pitch")
    L = load(os.path.join(args.reference, "app", "legacy", "panorama_to_plane.py"), "ref_legacy")

    arrays, meta = {}, {"cv2": cv2.__version__, "numpy": np.__version__, "reference_version": P.get_version(),
                        "views": [], "legacy": []}
    for (name, pw, ph, seed, kind, yaws, pitches, ow, oh, fov) in VIEW_CASES:
        pano = synth.synth_pano(pw, ph, seed, kind)
        arrays[name + "_pano"] = pano
        for yaw in yaws:
            P.yaw_mapping_cache.clear()
            views = P.process_yaw_and_pitchs(pano, yaw, pitches, ow, oh, fov)
            arrays["%s_y%d" % (name, yaw)] = np.stack(views)
        # the float maps the reference used, so that a failing comparison can tell map differences from gather ones
        for pitch in pitches:
            U, V = P.get_pitch_mapping(ow, oh, pitch, pw, ph, fov)
            arrays["%s_p%d_U" % (name, pitch)], arrays["%s_p%d_V" % (name, pitch)] = U, V
        meta["views"].append(dict(name=name, pw=pw, ph=ph, seed=seed, kind=kind, yaws=yaws, pitches=pitches, ow=ow, oh=oh, fov=fov))
    for (name, pw, ph, seed, yaw, pitch, fov, ow, oh) in LEGACY_CASES:
        pano = synth.synth_pano(pw, ph, seed, "N")
        U, V = L.precompute_mapping(ow, oh, np.radians(fov), np.radians(yaw), np.radians(pitch), pw, ph)
        arrays[name + "_pano"], arrays[name + "_U"], arrays[name + "_V"] = pano, U, V
        arrays[name + "_bilinear"] = L.panorama_to_plane(pano, U, V)
        for method in ("nearest", "bicubic"):
            arrays[name + "_" + method] = L.interpolate_color(U, V, pano, method=method)
        meta["legacy"].append(dict(name=name, pw=pw, ph=ph, seed=seed, yaw=yaw, pitch=pitch, fov=fov, ow=ow, oh=oh))
    meta["sha256"] = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in arrays.items()}
    arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(out, **arrays)
    print("wrote %s (%d arrays, cv2 %s)" % (out, len(arrays), cv2.__version__))


if __name__ == "__main__":
    main()
