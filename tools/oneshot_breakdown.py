#!/usr/bin/env python3
"""PCIe-inclusive cost of the host-buffer (one-shot) entry point on BASELINE config 2, split into its
parts: upload, kernel, download -- from pageable and from page-locked (p2p_host_alloc) host memory, with
and without the per-thread one-shot cache.  Prints one JSON object."""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd")
nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")

PW, PH, OW, OH = 8192, 4096, 1920, 1080
YAWS, PITCHES = list(range(0, 360, 30)), [60, 90, 120]


def best(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3


def main():
    pano = synth.synth_pano(PW, PH, 1000, "S")
    pin = nat.pinned_empty(pano.shape)
    pin[...] = pano
    out = {"workload": "cfg2: 8192x4096 -> 1920x1080, 12 yaw x 3 pitch", "unit": "ms (min, median of 5)",
           "bytes_up": int(pano.nbytes), "bytes_down": int(len(YAWS) * len(PITCHES) * OW * OH * 3)}
    npix = len(YAWS) * len(PITCHES) * OW * OH

    os.environ["P2P_ONESHOT_CACHE"] = "0"
    nat.reload_options()
    nat.remap_views(pano, YAWS, PITCHES, 90, OW, OH)
    out["oneshot_pageable_nocache"] = best(lambda: nat.remap_views(pano, YAWS, PITCHES, 90, OW, OH))
    os.environ["P2P_ONESHOT_CACHE"] = "1"
    nat.reload_options()
    nat.remap_views(pano, YAWS, PITCHES, 90, OW, OH)
    out["oneshot_pageable_cached"] = best(lambda: nat.remap_views(pano, YAWS, PITCHES, 90, OW, OH))
    nat.remap_views(pin, YAWS, PITCHES, 90, OW, OH, pinned=True)
    out["oneshot_pinned_cached"] = best(lambda: nat.remap_views(pin, YAWS, PITCHES, 90, OW, OH, pinned=True))
    out["oneshot_pinned_cached_new_yaws"] = best(
        lambda: nat.remap_views(pin, [(y + int(time.perf_counter() * 1e6) % 29 + 1) % 360 for y in YAWS], PITCHES, 90, OW, OH, pinned=True))
    nat.release_cache()

    ctx = nat.Context(0)
    job = nat.Job(ctx, PW, PH, 1, YAWS, PITCHES, 90, OW, OH)
    job.set_pano(0, pano)
    job.run()
    ctx.synchronize()
    out["upload_pageable"] = best(lambda: job.set_pano(0, pano))
    out["upload_pinned"] = best(lambda: job.set_pano(0, pin))
    out["kernel"] = best(lambda: (job.run(), ctx.synchronize()))
    out["download_pageable"] = best(lambda: job.get_views())
    job.get_views(pinned=True)
    out["download_pinned"] = best(lambda: job.get_views(pinned=True))
    out["pinned_alloc_224MB_uncached"] = None
    nat._pool.trim()
    t0 = time.perf_counter()
    a = nat.pinned_empty((out["bytes_down"],))
    out["pinned_alloc_224MB_uncached"] = (time.perf_counter() - t0) * 1e3
    del a
    for k in ("oneshot_pageable_nocache", "oneshot_pageable_cached", "oneshot_pinned_cached"):
        out[k + "_Gpix_s"] = npix / out[k][0] / 1e6
    out["upload_pinned_GBs"] = out["bytes_up"] / out["upload_pinned"][0] / 1e6
    out["download_pinned_GBs"] = out["bytes_down"] / out["download_pinned"][0] / 1e6
    out["upload_pageable_GBs"] = out["bytes_up"] / out["upload_pageable"][0] / 1e6
    out["download_pageable_GBs"] = out["bytes_down"] / out["download_pageable"][0] / 1e6
    job.close()
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
