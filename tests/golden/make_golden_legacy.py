#!/usr/bin/env python3
"""Generate tests/golden/legacy_maps_golden.npz by IMPORTING THE REFERENCE's legacy map builder.

Build container only (needs /root/reference):   python tests/golden/make_golden_legacy.py

app/legacy/panorama_to_plane.py imports cv2 at import time (L:3); an empty stub module stands in for it.
Only the NumPy-only functions run: get_rotation_matrix (L:21-45), precompute_mapping (L:47-157),
check_pitch (L:196-216), check_yaw (L:218-237).  The output is DATA: inputs and the reference's outputs.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/app/legacy/panorama_to_plane.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "legacy_maps_golden.npz")


def load_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("ref_legacy_panorama_to_plane", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    warnings.simplefilter("ignore")
    ref = load_reference()
    arrays, meta = {}, {"numpy": np.__version__, "tiny": [], "sampled": [], "rot": [], "check_yaw": [], "check_pitch": []}
    for yaw, pitch in ((0, 90), (30, 90), (77, 60), (180, 120), (300, 30), (359, 150), (360, 1), (45, 179)):
        R = ref.get_rotation_matrix(np.radians(yaw), np.radians(pitch))
        key = f"rot_y{yaw}_p{pitch}"
        arrays[key] = R
        meta["rot"].append({"key": key, "yaw": yaw, "pitch": pitch})
        for fov in (60, 90):
            key = f"tiny_f{fov}_y{yaw}_p{pitch}"
            U, V = ref.precompute_mapping(64, 48, float(np.radians(fov)), float(np.radians(yaw)),
                                          float(np.radians(pitch)), 256, 128)
            arrays[key + "_U"], arrays[key + "_V"] = U, V
            meta["tiny"].append({"key": key, "W": 64, "H": 48, "fov": fov, "yaw": yaw, "pitch": pitch, "pw": 256, "ph": 128})
    # the legacy CLI's defaults (L:286-289): 1000x1500, FOV 90, pitch 90, yaws 0..300 step 60, on an 8K panorama
    for yaw in (0, 60, 120, 240):
        U, V = ref.precompute_mapping(1000, 1500, float(np.radians(90)), float(np.radians(yaw)), float(np.radians(90)), 8192, 4096)
        key = f"cli_y{yaw}"
        arrays[key + "_U"], arrays[key + "_V"] = U[::25, ::20].copy(), V[::25, ::20].copy()
        meta["sampled"].append({"key": key, "W": 1000, "H": 1500, "fov": 90, "yaw": yaw, "pitch": 90, "pw": 8192, "ph": 4096,
                                "stride_y": 25, "stride_x": 20, "sha_U": sha(U), "sha_V": sha(V),
                                "nan_U": int(np.isnan(U).sum()), "nan_V": int(np.isnan(V).sum())})
    for vals in ([0, 60, 120], [300, 0, 0, 60], [360], [361], [-1, 5]):
        try:
            meta["check_yaw"].append({"in": vals, "out": ref.check_yaw(list(vals))})
        except argparse.ArgumentTypeError as e:
            meta["check_yaw"].append({"in": vals, "error": str(e)})
    for val in ("1", "90", "179", "0", "180", "abc", "45.5"):
        try:
            meta["check_pitch"].append({"in": val, "out": ref.check_pitch(val)})
        except argparse.ArgumentTypeError as e:
            meta["check_pitch"].append({"in": val, "error": str(e)})
    arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **arrays)
    print(OUT, os.path.getsize(OUT), "bytes;", len(arrays), "arrays")


if __name__ == "__main__":
    main()
