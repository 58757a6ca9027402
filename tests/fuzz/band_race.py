#!/usr/bin/env python3
"""Repeatability of the band kernel (and, at the end, of the other view kernels): the reference CLI's default view set at 8K drawn again and again, cold (a new context
and plan every time) and warm (one job, many launches), every result against the per-view tiles' bytes.
GPU box, repo root:   python3 tests/fuzz/band_race.py [cold runs] [warm runs]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")

PW, PH, OW, OH, FOV = 8192, 4096, 800, 800, 90
YAWS, PITCHES = [0, 90, 180, 270], [30, 60, 90, 120, 150]


def job_of(ctx, pano):
    job = nat.Job(ctx, PW, PH, 1, YAWS, PITCHES, FOV, OW, OH)
    job.set_pano(0, pano)
    return job


def main():
    cold = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    warm = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    pano = synth.synth_pano(PW, PH, 4242, "N")
    os.environ["P2P_BAND"] = "0"; nat.reload_options()
    ctx = nat.Context(0); job = job_of(ctx, pano); job.run(); ctx.synchronize()
    want = job.get_views(0).copy(); job.close(); ctx.close()
    os.environ["P2P_BAND"] = "-1"; nat.reload_options()
    bad_cold = bad_warm = 0
    where = {}
    for i in range(cold):
        ctx = nat.Context(0); job = job_of(ctx, pano); job.run(); ctx.synchronize()
        got = job.get_views(0)
        d = np.argwhere((got != want).any(axis=-1))
        if len(d):
            bad_cold += 1
            for r in d[:6]:
                where[tuple(int(x) for x in r)] = where.get(tuple(int(x) for x in r), 0) + 1
        job.close(); ctx.close()
    ctx = nat.Context(0); job = job_of(ctx, pano)
    info = None
    for i in range(warm):
        job.run(); ctx.synchronize()
        got = job.get_views(0)
        d = np.argwhere((got != want).any(axis=-1))
        if len(d):
            bad_warm += 1
            for r in d[:6]:
                where[tuple(int(x) for x in r)] = where.get(tuple(int(x) for x in r), 0) + 1
    info = job.info()
    job.close(); ctx.close()
    for k, v in sorted(where.items())[:30]:
        print("   (yaw, pitch, row, col) = %s: %d times" % (k, v))
    # the other kernels the same way, each against its own first launch: config 2 through the per-view tiles of either
    # shape, the CLI set through the gather kernel
    bad_other = 0
    for name, env, geo in (("config 2, 64-wide tiles", {}, (8192, 4096, list(range(0, 360, 30)), [60, 90, 120], 1920, 1080)),
                           ("config 2, 128-wide tiles", {"P2P_TILE_SHAPE": "128"}, (8192, 4096, list(range(0, 360, 30)), [60, 90, 120], 1920, 1080)),
                           ("CLI set, per-view tiles", {"P2P_BAND": "0"}, (PW, PH, YAWS, PITCHES, OW, OH))):
        os.environ.update(env); nat.reload_options()
        pw, ph, yaws, pitches, ow, oh = geo
        ctx = nat.Context(0); job = nat.Job(ctx, pw, ph, 1, yaws, pitches, FOV, ow, oh); job.set_pano(0, pano)
        job.run(); ctx.synchronize(); first = job.get_views(0).copy()
        n_bad = 0
        for i in range(max(1, warm // 6)):
            job.run(); ctx.synchronize()
            n_bad += int(not np.array_equal(job.get_views(0), first))
        job.close(); ctx.close()
        for k in env:
            os.environ.pop(k, None)
        nat.reload_options()
        print("%s: launches that differ from the first %d of %d" % (name, n_bad, max(1, warm // 6)))
        bad_other += n_bad
    print("band tiles %d: cold runs with wrong pixels %d of %d, warm %d of %d; other kernels: %d launches differ" %
          (info["band_tiles"], bad_cold, cold, bad_warm, warm, bad_other))
    return 1 if bad_cold or bad_warm or bad_other else 0


if __name__ == "__main__":
    sys.exit(main())
