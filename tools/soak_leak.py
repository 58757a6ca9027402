#!/usr/bin/env python3
"""Soak: many images of changing geometry through every host-side path (one-shot calls from several threads, the
two-slot device pipeline, view-sharded runs, job create / close) while watching free device memory and the
process's resident set: neither may drift.  Diagnostic; run on the GPU box:  python tools/soak_leak.py [rounds]"""
import importlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
drv = importlib.import_module("360-to-planer-images_amd._driver")
synth = importlib.import_module("360-to-planer-images_amd.synth")
import argparse
_ap = argparse.ArgumentParser(description=__doc__, allow_abbrev=False)
_ap.add_argument("--rounds", type=int, default=30)
rounds = _ap.parse_args().rounds
if not 1 <= rounds <= 1000:
    sys.exit("--rounds must be in 1..1000")


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0
    return 0.0


def free_mb():
    # (every call below returns with its device work done; the driver's own figure, no second runtime in the process)
    return nat.device_mem_info(0)[0] / 2**20


panos = {(2048, 1024): synth.synth_pano(2048, 1024, 1, "N"), (4096, 2048): synth.synth_pano(4096, 2048, 2, "S"),
         (1000, 500): synth.synth_pano(1000, 500, 3, "N")}
geoms = [((2048, 1024), [0, 90, 181.5], [60, 90], 90, 640, 360), ((4096, 2048), [0, 30, 60, 90], [45, 90, 135], 75, 800, 800),
         ((1000, 500), [10, 20], [90], 100, 333, 200), ((2048, 1024), list(range(0, 360, 45)), [30, 150], 60, 512, 512)]
log = []
for r in range(rounds):
    for (pk, yaws, pitches, fov, ow, oh) in geoms:
        pano = panos[pk]
        # one-shot entry point from 4 threads at once
        ths = [threading.Thread(target=lambda: nat.remap_views_f64(pano, yaws, pitches, fov, ow, oh)) for _ in range(4)]
        [t.start() for t in ths]; [t.join() for t in ths]
        # pipeline: 6 images through two slots, then closed
        pipe = drv.DevicePipeline(0)
        tickets = [pipe.submit(pano, yaws, pitches, fov, ow, oh) for _ in range(6)]
        [t.result() for t in tickets]
        pipe.close()
        # view-sharded over three contexts on device 0
        drv.process_views_sharded(pano, yaws, pitches, ow, oh, fov, [0, 0, 0])
        # float path job, created and closed
        nat.remap_views_f64(pano, yaws, pitches, fov, ow, oh, flags=nat.FLAG_PIXELS_F16)
    log.append((r, free_mb(), rss_mb()))
    if r % 5 == 0 or r == rounds - 1:
        print("round %3d  free device memory %9.1f MB   host RSS %8.1f MB" % log[-1], flush=True)
settle = max(2, rounds // 3)   # caches (one-shot slots, pinned pool, shared contexts) fill during the first rounds
d_free = log[-1][1] - log[settle][1]
d_rss = log[-1][2] - log[settle][2]
print("drift after round %d: device %+.1f MB, host RSS %+.1f MB" % (settle, d_free, d_rss))
sys.exit(0 if (d_free > -64 and d_rss < 256) else 1)
