#!/usr/bin/env python3
"""Generate tests/golden/maps_golden.npz by IMPORTING THE REFERENCE's own map builders.

Run in the build container only (needs /root/reference; the GPU box has none):

    python tests/golden/make_golden_maps.py

The reference module does `import cv2` at import time (P:8) and cv2 is not installable
here, so an empty stub module named cv2 is placed in sys.modules first.  Only the
NumPy-only functions are called: precompute_yaw_mapping (P:79-108),
get_pitch_mapping / precompute_pitch_mapping (P:55-73, P:114-175), check_pitch
(P:362-376), get_version (P:22-27).  cv2.remap is never reached.

The output is DATA (inputs + the reference's outputs); no reference source is stored.
"""
import hashlib
import importlib.util
import json
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/app/panorama_to_plane-pitch.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "maps_golden.npz")


def load_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("ref_panorama_to_plane_pitch", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    warnings.simplefilter("ignore")
    ref = load_reference()
    arrays = {}
    meta = {
        "numpy": np.__version__,
        "reference_version": ref.get_version(),
        "avx512f": "avx512f" in open("/proc/cpuinfo").read(),
        "tiny": [], "yaw_tables": [], "sampled": [], "nan_pixels": [], "known": [],
    }

    # G1: full (U, V) for tiny configs
    for fov in (60, 90, 120):
        for pitch in (1, 30, 45, 90, 135, 179):
            key = f"tiny_f{fov}_p{pitch}"
            U, V = ref.get_pitch_mapping(64, 48, pitch, 256, 128, fov)
            arrays[key + "_U"], arrays[key + "_V"] = U, V
            meta["tiny"].append({"key": key, "ow": 64, "oh": 48, "pitch": pitch, "pw": 256, "ph": 128, "fov": fov})
    for yaw in (0, 1, 30, 45, 77, 90, 359, 360, -30, 400):
        key = f"tinyyaw_{yaw}"
        U, V = ref.precompute_yaw_mapping(256, 8, yaw)
        arrays[key + "_U"], arrays[key + "_V"] = U, V
        meta["tiny"].append({"key": key, "pw": 256, "ph": 8, "yaw": yaw})

    # G3: yaw column tables U[0, :] (rows are identical; the generator checks it)
    for pw in (2048, 8192, 16384):
        for yaw in (0, 1, 30, 45, 77, 90, 359, 360, -30, 400):
            U, V = ref.precompute_yaw_mapping(pw, 4, yaw)
            assert (U == U[0]).all() and (V == np.arange(4, dtype=np.float32)[:, None]).all()
            key = f"yawtab_{pw}_{yaw}"
            arrays[key] = U[0].copy()
            meta["yaw_tables"].append({"key": key, "pw": pw, "yaw": yaw})

    # G2/G4: strided samples + sha256 of full maps for the BASELINE configs
    big = [
        # (ow, oh, pitch, pw, ph, fov, stride)
        (512, 512, 90, 2048, 1024, 90, 8),     # cfg 1
        (1920, 1080, 60, 8192, 4096, 90, 16),  # cfg 2
        (1920, 1080, 90, 8192, 4096, 90, 16),
        (1920, 1080, 120, 8192, 4096, 90, 16),
        (800, 800, 30, 4096, 2048, 90, 16),    # reference CLI defaults (P:412-430)
        (800, 800, 150, 4096, 2048, 90, 16),
        (4096, 4096, 30, 16384, 8192, 60, 64),  # cfg 4 (pole in view)
        (4096, 4096, 90, 16384, 8192, 60, 64),
        (4096, 4096, 150, 16384, 8192, 60, 64),
    ]
    for ow, oh, pitch, pw, ph, fov, st in big:
        U, V = ref.get_pitch_mapping(ow, oh, pitch, pw, ph, fov)
        key = f"samp_{ow}x{oh}_p{pitch}_{pw}x{ph}_f{fov}"
        arrays[key + "_U"] = U[::st, ::st].copy()
        arrays[key + "_V"] = V[::st, ::st].copy()
        meta["sampled"].append({"key": key, "ow": ow, "oh": oh, "pitch": pitch, "pw": pw, "ph": ph,
                                "fov": fov, "stride": st, "sha_U": sha(U), "sha_V": sha(V)})
        ref.pitch_mapping_cache.clear()

    # G5: NaN pixels (arccos of a value rounded above 1, P:162)
    for ow, oh, fov, pitches in ((1920, 1080, 90, (5, 12, 168, 175, 90)), (800, 800, 90, (4, 5, 175, 176, 90))):
        for pitch in pitches:
            U, V = ref.get_pitch_mapping(ow, oh, pitch, 8192, 4096, fov)
            meta["nan_pixels"].append({"ow": ow, "oh": oh, "pitch": pitch, "pw": 8192, "ph": 4096, "fov": fov,
                                       "nan_V": np.argwhere(np.isnan(V)).tolist(),
                                       "nan_U": int(np.isnan(U).sum())})
            ref.pitch_mapping_cache.clear()

    # G6: geometry known answers (view centre, wrap column, clamp value)
    U, V = ref.get_pitch_mapping(512, 512, 90, 2048, 1024, 90)
    meta["known"].append({"what": "centre_512_p90", "U": float(U[256, 256]), "V": float(V[256, 256])})
    U, V = ref.get_pitch_mapping(1920, 1080, 60, 8192, 4096, 90)
    meta["known"].append({"what": "centre_1080p_p60", "U": float(U[540, 960]), "V": float(V[540, 960])})
    U, _ = ref.precompute_yaw_mapping(2048, 2, 30)
    meta["known"].append({"what": "yaw30_2048_cols_1876_1879", "U": [float(x) for x in U[0, 1876:1880]]})

    # validators (P:362-376)
    cp = {}
    for s in ("1", "179", "0", "180", "90", "abc", "-5", "45.5"):
        try:
            cp[s] = ref.check_pitch(s)
        except Exception as e:  # argparse.ArgumentTypeError
            cp[s] = "ERR:" + type(e).__name__
    meta["check_pitch"] = cp

    arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, os.path.getsize(OUT) / 1e6, "MB")


if __name__ == "__main__":
    main()
