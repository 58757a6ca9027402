#!/usr/bin/env python3
"""Wall time of the map getters of the drop-in module (device kernels + copy back to NumPy arrays), for the sizes
SURVEY 3.5 quotes the reference's NumPy builders on (9.5 s per new yaw at 8K, 0.27 s per new pitch at 1080p,
6.9 s at 4096^2)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd")
P = pkg.panorama_to_plane_pitch


def best(f, n=3):
    ts = []
    for _ in range(n):
        P.yaw_mapping_cache.clear(); P.pitch_mapping_cache.clear()
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3


pkg.get_pitch_mapping(64, 64, 90, 256, 128)
print("get_yaw_mapping(8192, 4096, 30)                      %8.1f ms" % best(lambda: pkg.get_yaw_mapping(8192, 4096, 30)))
print("get_pitch_mapping(1920, 1080, 60, 8192, 4096)        %8.1f ms" % best(lambda: pkg.get_pitch_mapping(1920, 1080, 60, 8192, 4096)))
print("get_pitch_mapping(4096, 4096, 30, 16384, 8192, 60)   %8.1f ms" % best(lambda: pkg.get_pitch_mapping(4096, 4096, 30, 16384, 8192, 60)))
legacy = importlib.import_module("360-to-planer-images_amd.panorama_to_plane")
import numpy as np
def leg():
    legacy.precompute_mapping.cache_clear()
    legacy.precompute_mapping(1000, 1500, float(np.radians(90)), float(np.radians(60)), float(np.radians(90)), 8192, 4096)
print("legacy precompute_mapping(1000, 1500, ..., 8192, 4096) %6.1f ms" % best(leg))
