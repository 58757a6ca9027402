#!/usr/bin/env python3
"""Timing probes of remap_views_kernel on the config-2 geometry: cost per pitch angle (footprint size, tiles that
leave the LDS scheme), per yaw kind (copy / blend) and per yaw count (set-up amortisation).  Diagnostic."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
CFG4 = len(sys.argv) > 1 and sys.argv[1] == "cfg4"   # 16384 x 8192 -> 4096 x 4096, FOV 60: cost per pitch angle
PW, PH, OW, OH, FOV = (16384, 8192, int(os.environ.get("ABL_OW", 4096)), 4096, 60) if CFG4 else (8192, 4096, 1920, 1080, 90)
pano = synth.synth_pano(PW, PH, 1000, "S")
ctx = nat.Context(0)


def run(yaws, pitches, label, n=200):
    job = nat.Job(ctx, PW, PH, 1, yaws, pitches, FOV, OW, OH)
    job.set_pano(0, pano)
    job.time_launches(False)
    for _ in range(60):
        job.run()
    ctx.mark(0)
    for _ in range(n):
        job.run()
    ctx.mark(1)
    ms = ctx.marked_ms() / n
    npx = len(yaws) * len(pitches) * OW * OH
    print("%-44s %8.1f us  %7.1f Gpix/s" % (label, ms * 1e3, npx / ms / 1e6), flush=True)
    job.close()


Y12 = list(range(0, 360, 30))
if CFG4:
    Y72 = list(range(0, 360, 5))
    for p in [int(x) for x in sys.argv[2:] if x.isdigit()] or (30, 60, 90, 120, 150):
        run(Y72, [p], "cfg4: 72 yaws x pitch %d" % p, n=20)
    if len(sys.argv) > 2 and sys.argv[2] == "multi":
        run(Y72, [90], "cfg4: 72 yaws x pitch 90", n=10)
        run(Y72, [90, 90], "cfg4: 72 yaws x pitches 90 90", n=10)
        run(Y72, [90, 90, 90], "cfg4: 72 yaws x pitches 90 90 90", n=10)
        run(Y72, [90, 90, 90, 90, 90], "cfg4: 72 yaws x pitches 90 x5", n=10)
        run(Y72[:24], [90, 90, 90], "cfg4: 24 yaws x pitches 90 90 90", n=10)
        run(Y72[:24], [60, 90, 120], "cfg4: 24 yaws x pitches 60 90 120", n=10)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "size":
        for k in (1, 2, 3, 5):
            run([y / k for y in range(0, 360 * k, 5)], [90], "cfg4: %d yaws x pitch 90" % (72 * k), n=10)
        sys.exit(0)
    run(Y72, [30, 60, 90, 120, 150], "cfg4: 72 yaws x 5 pitches (the whole job)", n=10)
    run(Y72, [60, 90, 120], "cfg4: 72 yaws x pitches 60 90 120", n=10)
    run(Y72, [30, 150], "cfg4: 72 yaws x pitches 30 150", n=10)
elif len(sys.argv) > 1 and sys.argv[1] == "pitch":
    for p in [int(x) for x in sys.argv[2:]] or (56, 58, 60, 62, 64, 66, 70, 75, 80, 90):
        run(Y12, [p], "12 yaws x pitch %d" % p)
else:
    run(Y12, [60, 90, 120], "cfg2")
    run([0, 45, 90, 135, 180, 225, 270, 315, 0, 45, 90, 135], [60, 90, 120], "cfg2 all copy yaws (f=0)")
    run([7, 37, 67, 97, 127, 157, 187, 217, 247, 277, 307, 337], [60, 90, 120], "cfg2 all fractional yaws")
    for n in (1, 2, 4, 24, 36):
        run(list(range(0, 360, 360 // n)) if n > 1 else [0], [60, 90, 120], "%d yaws" % n)
