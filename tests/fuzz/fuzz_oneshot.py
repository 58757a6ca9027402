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
# Usage: python tests/fuzz/fuzz_oneshot.py --cases CALLS_PER_THREAD --threads N(<= 8) --seed S   (named options only:
# round 3 lost two GPU boxes to "fuzz_oneshot.py 100000 8106" -- a seed in the thread-count position)
import _args
_a = _args.parser(__doc__, cases=200, seed=1000, threads=4).parse_args()
n_calls, n_threads, seed0 = _a.cases, _a.threads, _a.seed
GEOMS = [(512, 256, 96, 64, 90), (1024, 512, 200, 120, 90), (512, 256, 96, 64, 60), (2048, 1024, 320, 200, 100)]
PANOS = {(pw, ph): [synth.synth_pano(pw, ph, 40 + i, "N") for i in range(3)] for pw, ph, *_ in GEOMS}
errors = []
lock = threading.Lock()


_tls = threading.local()


def truth(pano, yaws, pitches, fov, ow, oh, flags=0, maps_=None):
    """The same views from a fresh resident job -- on ONE context per thread, kept for the thread's life (a context
    per call, times the threads, is what took the GPU down in round 3)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        ctx = _tls.ctx = nat.Context(0)
    ph, pw = pano.shape[:2]
    job = nat.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh, flags=flags)
    try:
        job.set_pano(0, pano)
        if maps_ is not None:
            job.set_maps(*maps_)
        job.run()
        return job.get_views(0)
    finally:
        job.close()


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
    ctx = getattr(_tls, "ctx", None)
    if ctx is not None:
        ctx.close()


t0 = time.time()
ths = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
[t.start() for t in ths]
[t.join() for t in ths]
for e in errors[:10]:
    print("ERROR", e)
print("fuzz_oneshot finished: %d threads x %d calls, %d errors, %.0f s" % (n_threads, n_calls, len(errors), time.time() - t0))
sys.exit(1 if errors else 0)
