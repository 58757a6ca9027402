#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLI on a folder of synthetic panoramas (decode + H2D + kernel + D2H +
encode + write), with the reference's default view set (4 yaws x 5 pitches, 800x800, P:412-437)."""
import importlib, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image
synth = importlib.import_module("360-to-planer-images_amd.synth")
n, pw, ph = int(os.environ.get("E2E_N", "4")), 4096, 2048
with tempfile.TemporaryDirectory() as d:
    os.makedirs(os.path.join(d, "in"))
    for i in range(n):
        Image.fromarray(synth.synth_pano(pw, ph, 1000 + i, "S")[:, :, ::-1]).save(os.path.join(d, "in", f"pano{i}.png"), compress_level=1)
    for workers in (1, 16):
        out = os.path.join(d, f"out{workers}")
        t = time.perf_counter()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "360-to-planer-images_amd", "panorama_to_plane_pitch.py"),
                               "--input_path", os.path.join(d, "in"), "--output_path", out, "--num_workers", str(workers)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t
        files = os.listdir(out)
        print("num_workers %2d: %d panoramas %dx%d -> %d views 800x800 in %.2f s (%.1f Mpix/s end to end, incl. interpreter start)"
              % (workers, n, pw, ph, len(files), dt, len(files) * 0.64 / dt))
