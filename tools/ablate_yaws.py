#!/usr/bin/env python3
"""Timing probes of remap_views_kernel on the config-2 geometry: cost per pitch angle (footprint size, tiles that
leave the LDS scheme), per yaw kind (copy / blend) and per yaw count (set-up amortisation).  Diagnostic."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
pano = synth.synth_pano(8192, 4096, 1000, "S")
ctx = nat.Context(0)


def run(yaws, pitches, label, n=200):
    job = nat.Job(ctx, 8192, 4096, 1, yaws, pitches, 90, 1920, 1080)
    job.set_pano(0, pano)
    job.time_launches(False)
    for _ in range(60):
        job.run()
    ctx.mark(0)
    for _ in range(n):
        job.run()
    ctx.mark(1)
    ms = ctx.marked_ms() / n
    npx = len(yaws) * len(pitches) * 1920 * 1080
    print("%-44s %8.1f us  %7.1f Gpix/s" % (label, ms * 1e3, npx / ms / 1e6), flush=True)
    job.close()


Y12 = list(range(0, 360, 30))
if len(sys.argv) > 1 and sys.argv[1] == "pitch":
    for p in [int(x) for x in sys.argv[2:]] or (56, 58, 60, 62, 64, 66, 70, 75, 80, 90):
        run(Y12, [p], "12 yaws x pitch %d" % p)
else:
    run(Y12, [60, 90, 120], "cfg2")
    run([0, 45, 90, 135, 180, 225, 270, 315, 0, 45, 90, 135], [60, 90, 120], "cfg2 all copy yaws (f=0)")
    run([7, 37, 67, 97, 127, 157, 187, 217, 247, 277, 307, 337], [60, 90, 120], "cfg2 all fractional yaws")
    for n in (1, 2, 4, 24, 36):
        run(list(range(0, 360, 360 // n)) if n > 1 else [0], [60, 90, 120], "%d yaws" % n)
