#!/usr/bin/env python3
"""Kernel times of the legacy entry point panorama_to_plane(pano, U, V) (L:159-194, BORDER_REFLECT): run under
rocprofv3 --kernel-trace --stats.  8K panorama through pitch maps: 1080p and 4096x4096 views (LDS-scheme tiles: the main
kernel) and minifying 800x800 views incl. a pole (gather-scheme tiles: under a legacy border mode the table kernel)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd")
synth = importlib.import_module("360-to-planer-images_amd.synth")
pano = synth.synth_pano(8192, 4096, 1000, "S")
for (ow, oh, pitch) in ((1920, 1080, 60), (1920, 1080, 90), (4096, 4096, 90), (800, 800, 30), (800, 800, 90)):
    U, V = pkg.get_pitch_mapping(ow, oh, pitch, 8192, 4096, 90)
    for _ in range(5):
        out = pkg.panorama_to_plane(pano, U, V)
print("done", out.shape)
