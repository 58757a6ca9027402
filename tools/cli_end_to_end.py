#!/usr/bin/env python3
"""End-to-end timing of the drop-in tool on a folder of synthetic panoramas, with a per-stage breakdown
(VERDICT r05 item 3): decode / swap / H2D / kernel / D2H / encode / write.

  * the tool's main() is run in THIS process on a folder of N PNG panoramas with the reference's default view set
    (4 yaws x 5 pitches, 800 x 800, P:412-437), at 4096 x 2048 and 8192 x 4096 inputs, 1 and 16 workers; the host
    stages are the thread-seconds the tool itself records (panorama_to_plane_pitch.stage_seconds);
  * "swap" is the channel swap between decoder and kernel and between kernel and encoder: the file-to-file path has
    none since round 6 (it was 672 ms per 8K panorama + 34 ms per 1080p view);
  * the device stages (H2D, kernel, D2H) overlap in the tool's two-slot pipeline; they are measured here un-overlapped,
    synchronously, on one image of each size through the same library calls.

    python tools/cli_end_to_end.py            # E2E_N=8 images per size by default
"""
import importlib
import os
import sys
import tempfile
import time

os.environ.setdefault("TQDM_DISABLE", "1")  # (the tool's progress bars: one per image, not what is measured here)

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image  # noqa: E402

synth = importlib.import_module("360-to-planer-images_amd.synth")
tool = importlib.import_module("360-to-planer-images_amd.panorama_to_plane_pitch")
nat = importlib.import_module("360-to-planer-images_amd._native")
N = int(os.environ.get("E2E_N", "8"))
YAWS, PITCHES, OW, OH = [0, 90, 180, 270], [30, 60, 90, 120, 150], 800, 800


def device_stages(pano):
    """One image, synchronously: (H2D ms, first launch incl. plan ms, steady-state launch ms, D2H ms)."""
    ph, pw = pano.shape[:2]
    ctx = nat.Context(0)
    job = nat.Job(ctx, pw, ph, 1, YAWS, PITCHES, 90, OW, OH)
    pin = nat.pinned_empty(pano.shape)
    pin[...] = pano
    job.set_pano(0, pin)  # warm the pool
    t = time.perf_counter(); job.set_pano(0, pin); h2d = time.perf_counter() - t
    t = time.perf_counter(); job.run(); ctx.synchronize(); first = time.perf_counter() - t
    job.run(); ctx.synchronize()
    job.time_launches(32)
    for _ in range(32):
        job.run()
    ctx.synchronize()
    steady = float(np.median(job.kernel_ms_last(32)))
    out = nat.pinned_empty((len(YAWS), len(PITCHES), OH, OW, 3))
    nat.check(nat.lib().p2p_job_get_views(job._h, 0, out.ctypes.data))
    t = time.perf_counter(); nat.check(nat.lib().p2p_job_get_views(job._h, 0, out.ctypes.data)); d2h = time.perf_counter() - t
    job.close(); ctx.close()
    return h2d * 1e3, first * 1e3, steady, d2h * 1e3


def main():
    import logging
    logging.disable(logging.CRITICAL)
    print("tools/cli_end_to_end.py: %d PNG panoramas per size, the reference's default view set (4 yaws x 5 pitches, 800 x 800);" % N)
    print("host stages = thread-seconds summed over the worker threads; wall = the tool's main() start to finish, in this process")
    for pw, ph in ((4096, 2048), (8192, 4096)):
        with tempfile.TemporaryDirectory() as d:
            os.makedirs(os.path.join(d, "in"))
            panos = [synth.synth_pano(pw, ph, 1000 + i, "S") for i in range(N)]
            for i, p in enumerate(panos):
                Image.fromarray(p).save(os.path.join(d, "in", f"pano{i}.png"), compress_level=1)
            h2d, first, steady, d2h = device_stages(panos[0])
            del panos
            print("\n%d x %d inputs -- device stages of ONE image, un-overlapped: H2D %.2f ms, first launch (plan + kernels) %.2f ms, "
                  "steady-state kernels %.3f ms, D2H of 20 views %.2f ms" % (pw, ph, h2d, first, steady, d2h))
            for workers in (1, 16):
                out = os.path.join(d, f"out{workers}")
                tool.stage_seconds = {}
                t = time.perf_counter()
                tool.main(os.path.join(d, "in"), out, YAWS, PITCHES, OW, OH, num_workers=workers)
                dt = time.perf_counter() - t
                st, tool.stage_seconds = tool.stage_seconds, None
                n_views = len(os.listdir(out))
                print("  num_workers %2d: %d views in %.2f s wall = %.1f Mpix/s end to end | decode %.2f s, swap 0.00 s, to_pinned %.2f s, "
                      "device_wait %.2f s, encode %.2f s, write %.2f s (thread-seconds)"
                      % (workers, n_views, dt, n_views * OW * OH / 1e6 / dt, st.get("decode", 0.0), st.get("to_pinned", 0.0),
                         st.get("device_wait", 0.0), st.get("encode", 0.0), st.get("write", 0.0)))


if __name__ == "__main__":
    main()
