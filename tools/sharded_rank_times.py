#!/usr/bin/env python3
"""What the view-sharded multi-GPU path of ONE image costs each rank: config 2's 36 views dealt to WORLD ranks
(_driver.rank_view_set: one masked job per rank), every rank's job timed on THIS GPU, one after the other.  The
largest time is what an N-GPU run of one image would take per launch (kernel only; uploads and downloads aside).
    python tools/sharded_rank_times.py [--world 1 2 4 8] [--launches 600]"""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
drv = importlib.import_module("360-to-planer-images_amd._driver")
synth = importlib.import_module("360-to-planer-images_amd.synth")

ap = argparse.ArgumentParser(description=__doc__, allow_abbrev=False)
ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--launches", type=int, default=600)
ap.add_argument("--how", default="auto", choices=["auto", "blocks", "round_robin", "cost", "rows"],
                help="how the pitch-major view list is dealt; rows: a band of rows of EVERY view per rank (p2p_job_set_rows)")
a = ap.parse_args()
pw, ph, ow, oh, fov = 8192, 4096, 1920, 1080, 90
yaws, pitches = list(range(0, 360, 30)), [60, 90, 120]
pano = synth.synth_pano(pw, ph, 1000, "S")
ctx = nat.Context(0)
base = None
for world in a.world:
    times = []
    bands = drv.shard_rows(oh, world, pitches, fov, ow) if a.how == "rows" else None
    for rank in range(world):
        if bands is not None:
            r0, r1 = bands[rank]
            if r1 <= r0:
                times.append(0.0)
                continue
            job = nat.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
            job.set_rows(r0, r1)
        else:
            yi, pi, mask, mine = drv.rank_view_set(len(yaws), len(pitches), world, rank, a.how, pitches)
            if not mine:
                times.append(0.0)
                continue
            job = nat.Job(ctx, pw, ph, 1, [yaws[y] for y in yi], [pitches[p] for p in pi], fov, ow, oh)
            if not mask.all():
                job.set_view_mask(mask)
        job.set_pano(0, pano)
        for _ in range(a.launches // 3):
            job.run()
        ctx.mark(0)
        for _ in range(a.launches):
            job.run()
        ctx.mark(1)
        times.append(ctx.marked_ms() / a.launches * 1e3)
        job.close()
    worst = max(times)
    base = base or worst
    print(a.how + " world %d: %s per rank %s, us per launch %s -> slowest %.1f us, %.2f x the one-GPU launch" %
          (world, "rows" if bands is not None else "views",
           [b - a_ for a_, b in bands] if bands is not None else [len(drv.rank_view_set(len(yaws), len(pitches), world, r, a.how, pitches)[3]) for r in range(world)],
           ["%.1f" % t for t in times], worst, base / worst), flush=True)
ctx.close()
