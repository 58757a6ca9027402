import importlib, sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
pkg = importlib.import_module("360-to-planer-images_amd")
synth = importlib.import_module("360-to-planer-images_amd.synth")
nat = pkg._native
pw, ph, ow, oh, fov = 8192, 4096, 1920, 1080, 90
yaws, pitches = list(range(0, 360, 30)), [60, 90, 120]
base = synth.synth_pano(pw, ph, 1000, "N")
pin = nat.pinned_empty(base.shape); pin[...] = base
ctx = nat.Context(0)
jobs = [nat.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh) for _ in range(2)]
for j in jobs:
    j.set_pano(0, pin); j.run(); j.wait()
outs = [nat.pinned_empty((12, 3, oh, ow, 3)) for _ in range(2)]
N = 10
def t(f):
    t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) / N * 1e3
def up():
    for i in range(N): jobs[i % 2].set_pano(0, pin, wait=False)
    for j in jobs: j.wait()
def down():
    for i in range(N): jobs[i % 2].get_views_async(0, out=outs[i % 2])
    for j in jobs: j.wait()
def both():
    for i in range(N):
        jobs[i % 2].set_pano(0, pin, wait=False)
        jobs[(i + 1) % 2].get_views_async(0, out=outs[i % 2])
    for j in jobs: j.wait()
def full():
    for i in range(N):
        j = jobs[i % 2]
        j.wait()
        j.set_pano(0, pin, wait=False); j.run(); j.get_views_async(0, out=outs[i % 2])
    for j in jobs: j.wait()
def sync_up():
    for i in range(N): jobs[0].set_pano(0, pin)
def sync_down():
    for i in range(N): jobs[0].get_views(0, pinned=False) if False else nat.check(nat.lib().p2p_job_get_views(jobs[0]._h, 0, outs[0].ctypes.data))
for name, f in (("upload async", up), ("download async", down), ("both directions", both), ("full pipeline", full), ("sync upload", sync_up), ("sync download", sync_down)):
    f()
    print("%-18s %.2f ms / image" % (name, t(f)))
