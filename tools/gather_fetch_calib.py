#!/usr/bin/env python3
"""FETCH_SIZE calibration for the GATHER kernel's access pattern (12-byte loads from 4-byte-aligned addresses, a few
lanes per cache line): one 800 x 800 view of an 8K panorama (every tile gathers), the bytes its taps touch counted on
the host from the job's own coordinates -- unique 64-byte and 128-byte lines -- next to what rocprofv3 --pmc FETCH_SIZE
reports for the launch.  (The 1.92 factor of profiles/traffic.json was calibrated on the main kernel's 16-byte pieces.)
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/gather_fetch_calib.py [n_yaw]
prints the expected bytes; tools/pmc_summary.py OUT remap_views_gather_kernel gives the counter."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
n_yaw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
assert 1 <= n_yaw <= 4
pw, ph, ow, oh = 8192, 4096, 800, 800
pano = synth.synth_pano(pw, ph, 1000, "S")
ctx = nat.Context(0)
yaws = [0, 90, 180, 270][:n_yaw]                      # whole-column shifts on 8192 columns: stage 1 copies
job = nat.Job(ctx, pw, ph, 1, yaws, [90], 90, ow, oh)
job.set_pano(0, pano)
for _ in range(4):
    job.run()
ctx.synchronize()
c = job.get_coords()[0].astype(np.int64)               # (sx, sy) in 1/32 px
info = job.info()
ix, iy = c[..., 0] >> 5, c[..., 1] >> 5
src_pitch = (3 * (pw + 8) + 15) & ~15
lines64, lines128 = set(), set()
for y in yaws:
    s = (y * pw) // 360
    x = (ix + s) % pw
    a = (3 * x) & ~3
    for row in (np.clip(iy, 0, ph - 1), np.clip(iy + 1, 0, ph - 1)):
        base = row * src_pitch + a
        for off in (0, 11):
            lines64.update(np.unique((base + off) >> 6).tolist())
            lines128.update(np.unique((base + off) >> 7).tolist())
print("views %d, gather tiles %d of %d; the taps touch %.2f MB in 64-byte lines, %.2f MB in 128-byte lines (each fetched once)"
      % (n_yaw, info["n_gather_tiles"], info["n_tiles"], len(lines64) * 64 / 1e6, len(lines128) * 128 / 1e6), flush=True)
job.close(); ctx.close()
