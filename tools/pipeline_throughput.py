#!/usr/bin/env python3
"""Host-buffer throughput of the folder walk's device stage (decode and encode left out): config-2 images
(8192x4096 -> 36 views 1920x1080) through
  (a) the synchronous one-shot call per image (upload, kernel, download one after the other), and
  (b) _driver.DevicePipeline: two resident jobs, asynchronous copies on their own streams
      (upload of image k+1 and download of image k-1 under kernel k).
Run on the GPU box from the repo root: python tools/pipeline_throughput.py [n_images]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

pkg = importlib.import_module("360-to-planer-images_amd")
drv = importlib.import_module("360-to-planer-images_amd._driver")
synth = importlib.import_module("360-to-planer-images_amd.synth")
nat = pkg._native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
pw, ph, ow, oh, fov = 8192, 4096, 1920, 1080, 90
yaws, pitches = list(range(0, 360, 30)), [60, 90, 120]
base = synth.synth_pano(pw, ph, 1000, "N")
panos = []
for i in range(4):  # page-locked, as the tool's decoder produces them
    a = nat.pinned_empty(base.shape)
    a[...] = np.roll(base, 97 * i, axis=1)
    panos.append(a)
out_mb = len(yaws) * len(pitches) * ow * oh * 3 / 1e6
res = {"images": n, "upload_MB": base.nbytes / 1e6, "download_MB": out_mb}

pkg.process_views(panos[0], yaws, pitches, ow, oh, fov)  # plan, pinned pool, clocks
t0 = time.perf_counter()
for i in range(n):
    v = pkg.process_views(panos[i % 4], yaws, pitches, ow, oh, fov)
    del v
res["one_shot_ms_per_image"] = (time.perf_counter() - t0) / n * 1e3

pipe = drv.DevicePipeline(0)
warm = [pipe.submit(panos[i % 4], yaws, pitches, float(fov), ow, oh) for i in range(6)]  # fills the page-locked pool
for t in warm:
    t.result()
del warm
t0 = time.perf_counter()
tickets = []
for i in range(n):
    tickets.append(pipe.submit(panos[i % 4], yaws, pitches, float(fov), ow, oh))
    if len(tickets) > 2:
        tickets.pop(0).result()
for t in tickets:
    t.result()
res["pipeline_ms_per_image"] = (time.perf_counter() - t0) / n * 1e3
pipe.close()
res["pipeline_GBps_both_ways"] = (res["upload_MB"] + out_mb) / res["pipeline_ms_per_image"]
res["pipeline_Gpix_per_s"] = len(yaws) * len(pitches) * ow * oh / res["pipeline_ms_per_image"] / 1e6
print(json.dumps(res, indent=1))
