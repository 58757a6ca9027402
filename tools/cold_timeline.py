#!/usr/bin/env python3
"""Where does a cold image's device time go?  N first launches of config 2 through fresh contexts (the GPU kept busy in
between so that the clocks stay up), meant to run under `rocprofv3 --kernel-trace`; `--parse DIR` then prints, per cold
image, every kernel's start (us after the first one's start) and duration.
GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/cold_timeline.py 12 [--cli]
          python3 tools/cold_timeline.py --parse $OUT"""
import csv, glob, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a cold image starts at a yaw table kernel that follows a marker kernel (the warm job's view kernel)
    runs, cur = [], None
    for s, e, n in rows:
        short = n.split("(")[0].split("::")[-1][:40]
        if "yaw_table_kernel" in n or (cur is None and "yaw_desc" in n):
            if cur:
                runs.append(cur)
            cur = []
        if cur is not None:
            cur.append((s, e, short))
            if len(cur) > 16:
                runs.append(cur); cur = None
    if cur:
        runs.append(cur)
    for i, run in enumerate(runs):
        t0 = run[0][0]
        print("cold image %d:" % i)
        for s, e, n in run:
            print("   +%8.1f us  %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        return parse(sys.argv[2])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    cli = "--cli" in sys.argv  # the reference CLI's default view set (a band plan) instead of config 2
    pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
    synth = importlib.import_module("360-to-planer-images_amd.synth")
    pw, ph, ow, oh, fov = 8192, 4096, 1920, 1080, 90
    yaws, pitches = list(range(0, 360, 30)), [60, 90, 120]
    if cli:
        ow, oh, yaws, pitches = 800, 800, [0, 90, 180, 270], [30, 60, 90, 120, 150]
    pano = synth.synth_pano(pw, ph, 1, "S")
    warm_ctx = nat.Context(0)
    warm = nat.Job(warm_ctx, pw, ph, 1, yaws, pitches, fov, ow, oh); warm.set_pano(0, pano)
    for i in range(n):
        for _ in range(300):
            warm.run()
        ctx = nat.Context(0)
        warm_ctx.synchronize()
        job = nat.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
        job.set_pano(0, pano)
        ctx.mark(0); job.run(); ctx.mark(1)
        first = ctx.marked_ms() * 1e3
        ctx.mark(0); job.run(); ctx.mark(1)   # the second launch: the per-XCD lists, the pair-context table
        second = ctx.marked_ms() * 1e3
        ctx.mark(0); job.run(); ctx.mark(1)
        print("cold image %d: %.1f us (events); second launch %.1f us, third %.1f us" % (i, first, second, ctx.marked_ms() * 1e3), flush=True)
        ctx.synchronize()
        job.close(); ctx.close()
    warm.close(); warm_ctx.close()


if __name__ == "__main__":
    main()
