#!/usr/bin/env python3
"""Timing of the source-band tiles on config 2 and the CLI default set under P2P_BAND_* settings.
    python3 tools/band_sweep.py "BH=8,CW=8" "BH=16,CW=4" ...      (GPU box, repo root; 'off' = per-view tiles)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")

JOBS = {
    "cfg2": (8192, 4096, list(range(0, 360, 30)), [60, 90, 120], 90, 1920, 1080),
    "cli": (8192, 4096, [0, 90, 180, 270], [30, 60, 90, 120, 150], 90, 800, 800),
}
which = os.environ.get("SWEEP_JOBS", "cfg2,cli").split(",")
N = int(os.environ.get("SWEEP_N", "400"))
panos = {}
for spec in sys.argv[1:] or ["off"]:
    for k in list(os.environ):
        if k.startswith("P2P_BAND"):
            del os.environ[k]
    if spec == "off":
        os.environ["P2P_BAND"] = "0"
    else:
        os.environ["P2P_BAND"] = "1"
        for kv in spec.split(","):
            if kv:
                k, v = kv.split("=")
                os.environ[("P2P_BAND_" + k) if not k.startswith("P2P_") else k] = v
    nat.reload_options()
    line = "%-28s" % spec
    for name in which:
        pw, ph, yaws, pitches, fov, ow, oh = JOBS[name]
        if (pw, ph) not in panos:
            panos[(pw, ph)] = synth.synth_pano(pw, ph, 1000, "S")
        ctx = nat.Context(0)
        job = nat.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
        job.set_pano(0, panos[(pw, ph)])
        for _ in range(N // 3):
            job.run()
        ctx.mark(0)
        for _ in range(N):
            job.run()
        ctx.mark(1)
        us = ctx.marked_ms() / N * 1e3
        i = job.info()
        line += "  %s %7.1f us (tiles %d, gather %d)" % (name, us, i["band_tiles"], i["n_gather_tiles"])
        job.close(); ctx.close()
    print(line, flush=True)
