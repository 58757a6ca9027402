#!/usr/bin/env python3
"""Source-band tiles against the per-view tiles (and the oracle): the same job with P2P_BAND=0 and P2P_BAND=1 must give
the same bytes.  GPU box, repo root:   python3 tests/fuzz/band_check.py [--time] [--big]
Test infrastructure (it uses the oracle for one small case): lives under tests/."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")


def run(band, pw, ph, n_panos, yaws, pitches, fov, ow, oh, panos, mask=None, launches=0, env=None):
    os.environ["P2P_BAND"] = str(band)
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    nat.reload_options()
    ctx = nat.Context(0)
    job = nat.Job(ctx, pw, ph, n_panos, yaws, pitches, fov, ow, oh)
    for i, p in enumerate(panos):
        job.set_pano(i, p)
    if mask is not None:
        job.set_view_mask(mask)
    job.run()
    ctx.synchronize()
    info = job.info()
    out = [job.get_views(i).copy() for i in range(n_panos)]
    us = None
    if launches:
        for _ in range(launches // 3):
            job.run()
        ctx.mark(0)
        for _ in range(launches):
            job.run()
        ctx.mark(1)
        us = ctx.marked_ms() / launches * 1e3
    job.close(); ctx.close()
    for k in (env or {}):
        os.environ.pop(k, None)
    return out, info, us


def case(name, pw, ph, yaws, pitches, fov, ow, oh, n_panos=1, kind="N", mask=None, launches=0, env=None):
    panos = [synth.synth_pano(pw, ph, 1000 + i, kind) for i in range(n_panos)]
    a, ia, ta = run(0, pw, ph, n_panos, yaws, pitches, fov, ow, oh, panos, mask, launches)
    b, ib, tb = run(1, pw, ph, n_panos, yaws, pitches, fov, ow, oh, panos, mask, launches, env)
    bad = 0
    for x, y in zip(a, b):
        if mask is not None:
            m = np.asarray(mask, bool)
            x, y = x[m], y[m]
        bad += int((x != y).sum())
    t = "" if ta is None else "  %.1f -> %.1f us" % (ta, tb)
    print("%-34s band tiles %6d (classic %6d, gather %5d -> %5d): %s%s" %
          (name, ib["band_tiles"], ia["n_tiles"], ia["n_gather_tiles"], ib["n_gather_tiles"], "EQUAL" if bad == 0 else "%d bytes differ" % bad, t), flush=True)
    return bad


def main():
    timed = "--time" in sys.argv
    big = "--big" in sys.argv
    L = 300 if timed else 0
    bad = 0
    # the oracle, once (small): band tiles against the CPU restatement itself
    from _util import oracle_views
    pw, ph, ow, oh = 1024, 512, 256, 192
    yaws, pitches = [0, 30, 77], [60, 90]
    pano = synth.synth_pano(pw, ph, 7, "N")
    got, info, _ = run(1, pw, ph, 1, yaws, pitches, 90, ow, oh, [pano])
    want = oracle_views(pano, yaws, pitches, ow, oh, 90)
    d = int((got[0] != want).sum())
    print("oracle, 1024x512 -> 256x192, device maps on noise: %d bytes differ of %d (band tiles %d)" % (d, want.size, info["band_tiles"]), flush=True)
    bad += case("small 2048x1024 -> 480x270", 2048, 1024, [0, 30, 77, 180], [60, 90, 120], 90, 480, 270)
    bad += case("odd width 2048x1024 -> 333x211", 2048, 1024, [0, 45, 91], [45, 90], 90, 333, 211)
    bad += case("poles 2048x1024 -> 400x400", 2048, 1024, [0, 90, 200], [1, 30, 90, 150, 179], 100, 400, 400)
    bad += case("two panoramas 2048x1024", 2048, 1024, [0, 10, 20, 30, 40], [70, 110], 90, 640, 360, n_panos=2)
    bad += case("minifying 4096x2048 -> 200x200", 4096, 2048, [0, 90, 180, 270], [30, 60, 90, 120, 150], 90, 200, 200)
    bad += case("wide fov 2048x1024 -> 320x240", 2048, 1024, [0, 33], [90], 150, 320, 240)
    m = np.zeros((4, 3), np.uint8); m[0, 0] = m[1, 0] = m[2, 1] = m[3, 2] = 1
    bad += case("view mask 2048x1024 -> 480x270", 2048, 1024, [0, 30, 77, 180], [60, 90, 120], 90, 480, 270, mask=m)
    bad += case("70 yaws (2 chunks) 2048x1024", 2048, 1024, list(range(0, 350, 5)), [80, 100], 90, 320, 200)
    bad += case("w128 forced 2048x1024 -> 640x360", 2048, 1024, [0, 30, 77, 180], [60, 90, 120], 90, 640, 360, env={"P2P_TILE_SHAPE": 128})
    if big or timed:
        bad += case("config 2", 8192, 4096, list(range(0, 360, 30)), [60, 90, 120], 90, 1920, 1080, kind="S", launches=L)
        bad += case("config 2, noise", 8192, 4096, list(range(0, 360, 30)), [60, 90, 120], 90, 1920, 1080, kind="N", launches=L)
        bad += case("CLI defaults at 8K", 8192, 4096, [0, 90, 180, 270], [30, 60, 90, 120, 150], 90, 800, 800, kind="N", launches=L)
        bad += case("config 5, 45 yaws", 8192, 4096, list(range(0, 360, 8)), [90], 90, 1920, 1080, kind="N", launches=L)
        bad += case("16K -> 1024x576, 12 x 3", 16384, 8192, list(range(0, 360, 30)), [60, 90, 120], 90, 1024, 576, kind="N", launches=L)
        bad += case("16K -> 2048x1152, 12 x 3", 16384, 8192, list(range(0, 360, 30)), [60, 90, 120], 90, 2048, 1152, kind="N", launches=L)
    print("band_check: %s" % ("OK" if bad == 0 and d <= want.size // 1000 else "FAILED"), flush=True)
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
