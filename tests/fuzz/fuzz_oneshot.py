#!/usr/bin/env python3
"""One-off fuzz of the host-buffer (one-shot) entry points and their per-thread cache: random sequences of calls
over a few geometries (in-kernel maps / caller maps / float pixel paths / legacy remap, pinned or not, changing
yaws and panoramas), from several threads at once; every result is compared with a fresh resident job."""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import maps
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
# Usage: python tests/fuzz/fuzz_oneshot.py [calls per thread] [threads, at most 32] [seed]
# (the SECOND argument is the thread count, not a seed as in the other fuzzers: round 3 lost two GPU boxes to
# "fuzz_oneshot.py 100000 8106" and "fuzz_oneshot.py 1000 9406" -- nine thousand threads, each creating contexts and
# streams; hence the cap)
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
if not 1 <= n_threads <= 32:
    sys.exit("fuzz_oneshot: the second argument is the number of THREADS (1..32), got %d" % n_threads)
GEOMS = [(512, 256, 96, 64, 90), (1024, 512, 200, 120, 90), (512, 256, 96, 64, 60), (2048, 1024, 320, 200, 100)]
PANOS = {(pw, ph): [synth.synth_pano(pw, ph, 40 + i, "N") for i in range(3)] for pw, ph, *_ in GEOMS}
errors = []
lock = threading.Lock()


def truth(pano, yaws, pitches, fov, ow, oh, flags=0, maps_=None):
    ctx = nat.Context(0)
    ph, pw = pano.shape[:2]
    job = nat.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh, flags=flags)
    job.set_pano(0, pano)
    if maps_ is not None:
        job.set_maps(*maps_)
    job.run()
    out = job.get_views(0)
    job.close(); ctx.close()
    return out


def worker(tid):
    rng = np.random.default_rng(seed0 + tid)
    for call in range(n_calls):
        pw, ph, ow, oh, fov = GEOMS[int(rng.integers(0, len(GEOMS)))]
        pano = PANOS[(pw, ph)][int(rng.integers(0, 3))]
        n_yaw = int(rng.choice([1, 2, 5]))
        yaws = [int(v) for v in rng.integers(0, 360, size=n_yaw)]
        pitches = [[60, 90], [30, 150], [90]][int(rng.integers(0, 3))]
        kind = int(rng.integers(0, 5))
        pinned = bool(rng.integers(0, 2))
        try:
            if kind == 0:
                got = nat.remap_views(pano, yaws, pitches, fov, ow, oh, pinned=pinned)
                want = truth(pano, yaws, pitches, fov, ow, oh)
            elif kind == 1:
                rows = np.stack([maps.yaw_column_table(pw, y) for y in yaws])
                UV = [maps.pitch_map_deg(ow, oh, p, pw, ph, fov) for p in pitches]
                U, V = np.stack([u for u, _ in UV]), np.stack([v for _, v in UV])
                got = nat.remap_views_maps(pano, rows, U, V)
                want = truth(pano, yaws, pitches, fov, ow, oh, maps_=(rows, U, V))
            elif kind == 2:
                fl = [nat.FLAG_PIXELS_F16, nat.FLAG_PIXELS_F32][int(rng.integers(0, 2))]
                got = nat.remap_views(pano, yaws, pitches, fov, ow, oh, flags=fl, pinned=pinned)
                want = truth(pano, yaws, pitches, fov, ow, oh, flags=fl)
            elif kind == 3:
                U, V = maps.pitch_map_deg(ow, oh, pitches[0], pw, ph, fov)
                got = nat.remap_maps(pano, U, V, border=nat.BORDER_REFLECT)
                rows = np.stack([maps.yaw_column_table(pw, 0)])
                want = None  # checked against the views kernel with an identity yaw below
                ref = nat.remap_maps(pano, U, V, border=nat.BORDER_REFLECT)
                if not np.array_equal(got, ref):
                    raise AssertionError("legacy remap not repeatable")
            else:
                got = pkg.process_yaw_and_pitchs(pano, yaws[0], pitches, ow, oh, fov)
                want = list(truth(pano, [yaws[0]], pitches, fov, ow, oh)[0])
                got, want = np.stack(got), np.stack(want)
            if want is not None and not np.array_equal(np.asarray(got), np.asarray(want)):
                raise AssertionError("result differs from a fresh job")
        except Exception as e:  # noqa: BLE001
            with lock:
                errors.append((tid, call, kind, (pw, ph, ow, oh, fov), yaws, pitches, repr(e)))
        if rng.random() < 0.03:
            nat.release_cache()


t0 = time.time()
ths = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
[t.start() for t in ths]
[t.join() for t in ths]
for e in errors[:10]:
    print("ERROR", e)
print("fuzz_oneshot finished: %d threads x %d calls, %d errors, %.0f s" % (n_threads, n_calls, len(errors), time.time() - t0))
sys.exit(1 if errors else 0)
