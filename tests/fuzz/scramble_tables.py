#!/usr/bin/env python3
"""Robustness self-test: the view kernels on plan tables (and, every other case, yaw tables) that have been
overwritten with pseudo-random words (P2P_SCRAMBLE_PLAN, csrc/p2p_host_plan.cpp).  They must draw garbage and nothing
worse: no GPU fault, no hang, the process goes on and a clean job afterwards is byte-exact.  Covers the main, gather,
rest and table kernels (plain and flickering yaws, caller rows that are not a shift, poles, minifying views, odd
widths, the legacy tool's border modes, the float pixel path).  Run by hand on the GPU box (a hole in the range checks
would take the GPU down, so this is not part of pytest):
    python tests/fuzz/scramble_tables.py [cases] [seed]        (P2P_LIB_PATH selects the audit build)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["P2P_PLAN_CACHE"] = "0"   # private tables: nothing scrambled is shared or kept
from _util import oracle_maps, oracle_views  # noqa: E402

pkg = importlib.import_module("360-to-planer-images_amd")
nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
import _args  # named options with hard caps (tests/fuzz/_args.py)
_a = _args.parser(__doc__, cases=60, seed=1).parse_args()
cases, seed = _a.cases, _a.seed
rng = np.random.default_rng(seed)
audit_hits = ok_runs = 0
for case in range(cases):
    pw = int(rng.choice([256, 512, 1024, 2048, 4096])); ph = pw // 2
    ow, oh = int(rng.integers(16, 500)), int(rng.integers(16, 300))
    fov = int(rng.choice([40, 60, 90, 120]))
    yaws = [float(v) for v in rng.integers(0, 360, size=int(rng.integers(1, 20)))]
    if rng.random() < 0.5:
        yaws[0] = 14.0 if pw == 8192 else float(rng.uniform(0, 360))   # real-valued: fractional shifts, now and then a flickering one
    pitches = [float(v) for v in rng.integers(3, 178, size=int(rng.integers(1, 4)))]
    n_panos = int(rng.integers(1, 3))
    flags = int(rng.choice([0, 0, 0, nat.FLAG_PIXELS_F16, nat.FLAG_PIXELS_F32]))
    pano = synth.synth_pano(pw, ph, 77 + case, "N")
    os.environ["P2P_SCRAMBLE_PLAN"] = str((case + 1) | (1 << 29 if case % 3 == 0 else 0) | (1 << 30 if case % 2 else 0))
    nat.reload_options()   # the library reads its environment once per process, and again on request
    ctx = nat.Context(0)
    try:
        job = nat.Job(ctx, pw, ph, n_panos, yaws, pitches, fov, ow, oh, flags=flags)
        for i in range(n_panos):
            job.set_pano(i, pano)
        if flags == 0 and rng.random() < 0.4:   # caller maps, with yaw rows that are NOT a shift every other time
            rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
            if rng.random() < 0.5:
                rows = np.clip(rows[:, ::-1] * 0.7, 0, pw - 1).astype(np.float32)
            job.set_maps(rows, U, V)
        try:
            job.run()
            job.get_views(0)
            ok_runs += 1
        except nat.P2PError as e:
            if "AUDIT" not in str(e):
                raise
            audit_hits += 1
        job.close()
        # the legacy tool's border modes go through the table kernel
        if case % 4 == 0:
            Um = rng.uniform(-20, pw + 20, size=(oh, ow)).astype(np.float32)
            Vm = rng.uniform(-20, ph + 20, size=(oh, ow)).astype(np.float32)
            try:
                nat.remap_maps(pano, Um, Vm, border=int(rng.integers(1, 5)))
            except nat.P2PError as e:
                if "AUDIT" not in str(e):
                    raise
                audit_hits += 1
    finally:
        ctx.close()
    if case % 10 == 9:
        print("case %d: %d runs completed, %d audit records" % (case, ok_runs, audit_hits), flush=True)
# the GPU and the library are still in order: a clean job, byte for byte
os.environ.pop("P2P_SCRAMBLE_PLAN")
nat.reload_options()
pano = synth.synth_pano(1024, 512, 5, "N")
yaws, pitches = [0, 33, 90], [45, 90, 160]
rows, U, V = oracle_maps(yaws, pitches, 200, 144, 1024, 512, 90)
got = nat.remap_views_maps(pano, rows, U, V)
want = oracle_views(pano, yaws, pitches, 200, 144, 90)
assert np.array_equal(got, want), "the clean job after the scrambled ones is wrong"
print("scramble_tables seed %d: %d cases, %d runs completed on garbage tables, %d stopped by an audit record; the clean job afterwards is "
      "byte-exact; library %s" % (seed, cases, ok_runs, audit_hits, os.path.basename(nat.LIB_PATH)))
