#!/usr/bin/env python3
"""One-off fuzz of the integer path: random jobs (sizes, FOVs, yaw / pitch lists, several panoramas, odd widths)
through p2p_job_* with the oracle's float maps, every byte compared with the CPU restatement.
Usage: python tests/fuzz/fuzz_parity.py [n_cases] [seed] [only_case | -1] [big | real]
With "big": large panoramas, views towards the poles, wide FOVs -- footprints that outgrow the LDS buffers
(plan pass, sub-tiles, compacted item lists, direct gathers)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import oracle_maps, oracle_views
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
import _args  # named options with hard caps (tests/fuzz/_args.py)
_p = _args.parser(__doc__, cases=100, seed=2026)
_p.add_argument("--only", type=int, default=-1, help="re-run one case, checking every yaw")
_p.add_argument("--launches", type=int, default=2, help="launches per job, each checked: the first goes out in grid order behind the plan "
                "pass; from the second on the per-XCD lists, the pair-context table and the merged gather launch are in play")
_p.add_argument("--mode", choices=("default", "big", "real"), default="default",
                help="big: large geometries; real: real-valued yaw / pitch / FOV, pitch anywhere in [0, 180]")
_p.add_argument("--rows", action="store_true", help="after the launches: the first panorama inverted, a random band of tile rows "
                "(p2p_job_set_rows), one more launch -- the band's rows must be the oracle's for the NEW panorama, every other row the old one's")
_a = _p.parse_args()
n_cases, seed = _a.cases, _a.seed
only = _a.only if _a.only >= 0 else None
big, real = _a.mode == "big", _a.mode == "real"
ctx = nat.Context(0)
t0 = time.time(); bad = 0
for case in range(int(os.environ.get("FUZZ_FIRST", "0")), n_cases):  # FUZZ_FIRST: resume a long run
    if only is not None and case != only:
        continue
    rng = np.random.default_rng(seed * 100003 + case)
    pw = int(rng.choice([256, 500, 512, 1000, 1024, 2048, 2050, 4096]))
    ph = max(16, pw // int(rng.choice([2, 2, 2, 3, 4])))
    ow, oh = int(rng.integers(8, 420)), int(rng.integers(8, 300))
    fov = int(rng.choice([30, 60, 90, 90, 120, 150]))
    n_yaw = int(rng.integers(1, 20))
    yaws = [int(v) for v in rng.integers(-360, 720, size=n_yaw)]
    if rng.random() < 0.3:
        yaws[0] = 14   # per-column weights on 8192; harmless elsewhere
    pitches = [int(v) for v in rng.integers(1, 180, size=int(rng.integers(1, 4)))]
    n_panos = int(rng.integers(1, 4))
    if big:
        pw = int(rng.choice([4096, 8192])); ph = pw // 2
        ow, oh = int(rng.integers(200, 900)), int(rng.integers(100, 600))
        fov = int(rng.choice([60, 90, 120, 140]))
        n_yaw = int(rng.integers(1, 5)); yaws = [int(v) for v in rng.integers(0, 360, size=n_yaw)]
        pitches = [int(v) for v in rng.choice([3, 10, 20, 30, 45, 60, 120, 150, 170, 177], size=int(rng.integers(1, 3)))]
        n_panos = int(rng.integers(1, 3))
    if real:
        yaws = [float(v) for v in rng.uniform(-400, 800, size=n_yaw)]
        pitches = [float(v) for v in rng.uniform(0, 180, size=len(pitches))]
        fov = float(rng.uniform(15, 160))
    panos = [synth.synth_pano(pw, ph, 9000 + 7 * case + i, "N") for i in range(n_panos)]
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    job = nat.Job(ctx, pw, ph, n_panos, yaws, pitches, fov, ow, oh)
    for i, p in enumerate(panos):
        job.set_pano(i, p)
    job.set_maps(rows, U, V)
    check_yaws = sorted(set([0, n_yaw - 1, int(rng.integers(0, n_yaw))])) if only is None else list(range(n_yaw))
    if only is not None:
        print(dict(pw=pw, ph=ph, ow=ow, oh=oh, fov=fov, yaws=yaws, pitches=pitches, n_panos=n_panos))
    wants = {}
    for launch in range(max(1, min(4, _a.launches))):
        job.run()
        for i, p in enumerate(panos):
            got = job.get_views(i)
            for yi in check_yaws:
                if (i, yi) not in wants:
                    wants[(i, yi)] = oracle_views(p, [yaws[yi]], pitches, ow, oh, fov)[0]
                want = wants[(i, yi)]
                if not np.array_equal(got[yi], want):
                    bad += 1
                    print("MISMATCH", dict(case=case, launch=launch, pw=pw, ph=ph, ow=ow, oh=oh, fov=fov, yaw=yaws[yi], yaw_index=yi, n_yaw=n_yaw,
                                           pitches=pitches, pano=i, n=int((got[yi] != want).sum())), flush=True)
    if _a.rows and oh >= 16:
        n_tr = (oh + 15) // 16
        a_ = int(rng.integers(0, n_tr)); b_ = int(rng.integers(a_ + 1, n_tr + 1))
        r0, r1 = 16 * a_, min(oh, 16 * b_)
        inv = 255 - panos[0]
        job.set_pano(0, inv)
        job.set_rows(r0, r1)
        job.run()
        got = job.get_views(0)
        for yi in check_yaws:
            new = oracle_views(inv, [yaws[yi]], pitches, ow, oh, fov)[0]
            want = wants[(0, yi)].copy()
            want[:, r0:r1] = new[:, r0:r1]
            if not np.array_equal(got[yi], want):
                bad += 1
                print("MISMATCH (rows %d..%d)" % (r0, r1), dict(case=case, pw=pw, ph=ph, ow=ow, oh=oh, fov=fov, yaw=yaws[yi], pitches=pitches,
                                                               inside=int((got[yi][:, r0:r1] != want[:, r0:r1]).sum()), outside=int((got[yi] != want).sum())), flush=True)
    job.close()
    if case % 10 == 9:
        print("case %d done, %.0f s, mismatches %d" % (case + 1, time.time() - t0, bad), flush=True)
print("fuzz finished: %d cases, %d mismatching views, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
