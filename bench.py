#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: Mpix/s remapped, 8K equirect -> 1080p x 36 views.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg1|cfg4|cfg5] [--scaling weak|strong]

A "step" is one pass of the hot path over one batch of synthetic input: every (yaw, pitch) view of
the rank's resident panorama(s), i.e. one launch of remap_views_kernel.  Inputs are resident in HBM
before the timed region; outputs stay in HBM.  With N > 1 (launched by torch.distributed.run, one
rank per GPU) the (image x yaw x pitch) batch is dealt to the ranks at IMAGE granularity: by default
every rank draws the metric's own configuration -- one 8K panorama of its own x 36 views, N images
in all, "scaling": "weak" (per-GPU work fixed as N grows; the N = 1 line is the same workload).
--workload cfg3 deals config 3's 64 panoramas instead, 64 / N resident per GPU ("strong": the batch
is fixed); --scaling strong on a single-panorama workload deals a band of ROWS of every view to each
rank (--shard views: whole views, 36 on 8 GPUs = 3 to 5 each, one masked job per rank).  It is independent work, so there is NO data-path collective;
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.
value = pixels of the whole job / that time.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- algorithmic bytes of one launch / mean launch duration (HIP events around every
                  launch, on the stream the kernel runs on) against the 8 TB/s HBM peak
  cpu_baseline -- the CPU restatement of the reference path (oracle/, "port") timed on this box's
                  host cores on a bounded sample of the same workload (N = 1 only)
"""
import argparse
import csv
import glob
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "360-to-planer-images_amd"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(pw=8192, ph=4096, ow=1920, oh=1080, fov=90, yaws=list(range(0, 360, 30)), pitches=[60, 90, 120],
                 name="8192x4096 pano -> 1920x1080, FOV 90, 12 yaw x 3 pitch = 36 views"),
    # BASELINE.json configs[2]: 64 panoramas x the config-2 view set, sharded over the ranks (64 / N resident per GPU)
    "cfg3": dict(pw=8192, ph=4096, ow=1920, oh=1080, fov=90, yaws=list(range(0, 360, 30)), pitches=[60, 90, 120],
                 n_panos_total=64,
                 name="64 panos 8192x4096 -> 1920x1080 x 36 views each, panoramas dealt round-robin to the GPUs"),
    "cfg1": dict(pw=2048, ph=1024, ow=512, oh=512, fov=90, yaws=[0], pitches=[90],
                 name="2048x1024 pano -> one 512x512 view, FOV 90 yaw 0 pitch 90"),
    "cfg4": dict(pw=16384, ph=8192, ow=4096, oh=4096, fov=60, yaws=list(range(0, 360, 5)), pitches=[30, 60, 90, 120, 150],
                 name="16384x8192 pano -> 4096x4096, FOV 60, 72 yaw x 5 pitch = 360 views"),
    "cfg5": dict(pw=8192, ph=4096, ow=1920, oh=1080, fov=90, yaws=list(range(360)), pitches=[90],
                 name="8192x4096 pano -> 1920x1080, FOV 90, 360 yaw x 1 pitch = 360 views"),
    # the reference CLI's own defaults (P:412-437) on an 8K panorama: strongly minifying, two pole views
    "cli": dict(pw=8192, ph=4096, ow=800, oh=800, fov=90, yaws=[0, 90, 180, 270], pitches=[30, 60, 90, 120, 150],
                name="8192x4096 pano -> 800x800, FOV 90, yaw 0/90/180/270 x pitch 30/60/90/120/150 (the reference CLI's defaults)"),
}


def algorithmic_bytes(w, n_panos):
    """every source byte read once, every output byte written once; maps are computed."""
    views = n_panos * len(w["yaws"]) * len(w["pitches"])
    return 3 * w["pw"] * w["ph"] * n_panos + 3 * w["ow"] * w["oh"] * views


# ------------------------------------------------------------------------------------------------
# distributed scaffolding (importable: tests/test_distributed_cpu.py drives it with gloo, 2 ranks)
# ------------------------------------------------------------------------------------------------
class Dist:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.pg = None
        # P2P_BENCH_FORCE_PG=1: create the process group at world size 1 too (exercises RCCL init, barrier
        # and the max-reduction on a single-GPU box)
        if self.world > 1 or os.environ.get("P2P_BENCH_FORCE_PG") == "1":
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                import torch

                # P2P_BENCH_BACKEND=gloo: dry run of the N > 1 path where the ranks share one GPU (RCCL refuses that)
                backend = os.environ.get("P2P_BENCH_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                import torch

                torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.pg = dist
            self.backend = backend

    def barrier(self):
        if self.pg is not None:
            if self.backend == "nccl":
                self.pg.barrier(device_ids=[self.local_rank])  # this rank's GPU, explicitly
            else:
                self.pg.barrier()

    def max_over_ranks(self, x):
        if self.pg is None:
            return float(x)
        import torch

        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.pg is not None:
            self.pg.destroy_process_group()


def shard_round_robin(n_items, world, rank):
    """Items (panoramas, or views when there are fewer panoramas than GPUs) dealt round-robin -- the product's own
    dealing function (360-to-planer-images_amd/_driver.py), which needs no GPU to import."""
    return importlib.import_module(PKG + "._driver").shard_round_robin(n_items, world, rank)


def run_timed(step, device_sync, dist, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides;
    returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step()
    device_sync()
    dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    dist.barrier()
    t1 = time.perf_counter()
    return dist.max_over_ranks(t1 - t0)


def cpu_model():
    """The host CPU's model name (the CPU baseline states core count AND model)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine() or "unknown"


# ------------------------------------------------------------------------------------------------
def cpu_baseline(w, budget_s=12.0):
    """The oracle (CPU restatement of P:181-221 + cv2.remap arithmetic) on a bounded sample of the same
    workload: whole yaws (each = one full-panorama yaw remap + every pitch view, map building included).
    Two figures: one thread, and the reference's own parallelism -- one task per yaw on
    min(n_yaw, int(0.9 * cores)) threads (P:252-265, P:304-306; the C restatement releases the GIL as cv2 does).
    `value` / `cores` are the threaded run, the single-thread rate is reported next to them."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import cpu_ref

    synth = importlib.import_module(PKG + ".synth")
    pano = synth.synth_pano(w["pw"], w["ph"], 1000, "S")
    per_yaw = len(w["pitches"]) * w["ow"] * w["oh"]
    cache = {}
    done, t0 = 0, time.perf_counter()
    for yaw in w["yaws"]:
        cpu_ref.process_yaw_and_pitchs(pano, yaw, w["pitches"], w["ow"], w["oh"], w["fov"], _pitch_cache=cache)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt1 = time.perf_counter() - t0
    one = done * per_yaw / dt1 / 1e6

    cores = os.cpu_count() or 1
    threads = max(1, min(len(w["yaws"]), int(cores * 0.9)))
    # bounded: two yaws per thread (about 2 x the single-yaw time if the threads scale, 2 x threads x that if not)
    n_thr = 2 * threads
    yaws = (list(w["yaws"]) * ((n_thr + len(w["yaws"]) - 1) // len(w["yaws"])))[:n_thr]
    cache = {}
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda y: cpu_ref.process_yaw_and_pitchs(pano, y, w["pitches"], w["ow"], w["oh"], w["fov"],
                                                             _pitch_cache=cache), yaws))
    dtn = time.perf_counter() - t0
    return {
        "value": len(yaws) * per_yaw / dtn / 1e6, "unit": "Mpix/s", "cores": threads, "kind": "port",
        "cpu_model": cpu_model(), "host_cores": cores,
        "value_1core": one,
        "sample": "threaded: %d yaws x %d pitches on %d threads in %.1f s; single thread: %d of %d yaws in %.1f s; "
                  "map building included; host has %d cores" % (len(yaws), len(w["pitches"]), threads, dtn, done,
                                                                len(w["yaws"]), dt1, cores),
    }


def secondary_lines(nat, ctx, pano8k, device):
    """The other BASELINE configs and the reference CLI's default view set, each timed over a few back-to-back
    launches (HIP events on the job's stream around the region, inputs and outputs resident) after the headline
    region: {name: {workload, ms_per_launch, launches, Gpix_s, frac_of_hbm_peak, algorithmic_bytes, kernels}}.
    Panorama content does not enter the timing: config 3's share re-uses one 8K panorama for its 8 resident ones and
    config 4's 16K panorama is the 8K one tiled 2 x 2."""
    import numpy as np

    out = {}
    plans = [
        ("cfg3_share_8_panos", "cfg3", 8, 0, 60, "one GPU's share of config 3 at 8 GPUs: 8 panoramas resident, 288 views per launch"),
        ("cfg4", "cfg4", 1, 0, 5, None),
        ("cfg5_u8", "cfg5", 1, 0, 60, None),
        ("cfg5_f16_quality_mode", "cfg5", 1, nat.FLAG_PIXELS_F16, 60,
         "opt-in float pixel path (one float resample, not the reference's arithmetic): a quality mode, no throughput claim"),
        ("cli_default_view_set", "cli", 1, 0, 200, None),
    ]
    for name, wl, n_panos, flags, launches, note in plans:
        w = WORKLOADS[wl]
        try:
            pano = pano8k if w["pw"] == 8192 else np.ascontiguousarray(np.tile(pano8k, (w["ph"] // 4096, w["pw"] // 8192, 1)))
            job = nat.Job(ctx, w["pw"], w["ph"], n_panos, w["yaws"], w["pitches"], w["fov"], w["ow"], w["oh"], flags=flags)
            for i in range(n_panos):
                job.set_pano(i, pano)
            job.time_launches(1)   # (the plan pass is timed for a job that asks: plan_ms below)
            job.run()
            job.time_launches(False)
            for _ in range(max(2, launches // 4)):
                job.run()
            ctx.mark(0)
            for _ in range(launches):
                job.run()
            ctx.mark(1)
            ms = ctx.marked_ms() / launches
            plan_ms, tables_ms = job.plan_ms()
            views = n_panos * len(w["yaws"]) * len(w["pitches"])
            b_alg = algorithmic_bytes(w, n_panos)
            out[name] = {"workload": note or w["name"], "ms_per_launch": ms, "launches": launches,
                         "Gpix_s": views * w["ow"] * w["oh"] / ms / 1e6, "algorithmic_bytes": b_alg,
                         "frac_of_hbm_peak": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "plan_ms": plan_ms, "yaw_tables_ms": tables_ms,
                         "kernels": "float_views_kernel" if flags else
                                    ("remap_views_band_kernel (source-band tiles, %d; its first workgroups draw the %d gather tiles around the poles)"
                                     % (job.info()["band_tiles"], job.info()["n_gather_tiles"]) if job.info()["band_tiles"] > 0 else
                                     "remap_views_kernel + remap_views_gather_kernel (tiles that do not fit the LDS scheme)")}
            job.close()
            del pano
        except Exception as e:  # a secondary line never takes the headline down
            out[name] = {"error": repr(e)}
    return out


def host_to_host_lines(pkg, nat, drv, pano8k, device):
    """What a caller with HOST buffers gets (end-to-end, H2D of the source and D2H of the views stated
    separately) on the metric's configuration -- never `value`, which has everything resident:
      e2e_host_to_host   _driver.DevicePipeline, the folder walk's device stage (decode / encode left out): two resident
                         jobs used alternately, the upload of image k + 1 and the download of image k - 1 under kernel k;
      oneshot            one p2p_remap_views_u8 call per image from page-locked memory: upload, kernel, download in turn;
      legacy_reflect_3ch SURVEY 8(f)3: panorama_to_plane(pano, U, V), one cv2.remap with BORDER_REFLECT (L:179) -- the
                         3-channel table kernel -- for one 1080p view of the 8K panorama, host to host.
    Wall-clock on this process's thread; the copies' own rates come from event-timed copies of the same buffers."""
    import numpy as np

    w = WORKLOADS["cfg2"]
    yaws, pitches, fov, ow, oh = w["yaws"], w["pitches"], w["fov"], w["ow"], w["oh"]
    out = {}
    up_b = int(pano8k.nbytes)
    down_b = len(yaws) * len(pitches) * ow * oh * 3
    try:
        panos = []
        for i in range(2):  # page-locked, as the tool's decoder produces them
            a = nat.pinned_empty(pano8k.shape)
            a[...] = np.roll(pano8k, 97 * i, axis=1)
            panos.append(a)
        # the copies alone, each way (one job, synchronous calls)
        ctx = nat.Context(device)
        job = nat.Job(ctx, w["pw"], w["ph"], 1, yaws, pitches, fov, ow, oh)
        job.set_pano(0, panos[0]); job.run(); v = job.get_views(0, pinned=True); del v
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); job.set_pano(0, panos[1]); ts.append(time.perf_counter() - t0)
        h2d_ms = min(ts) * 1e3
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); v = job.get_views(0, pinned=True); ts.append(time.perf_counter() - t0); del v
        d2h_ms = min(ts) * 1e3
        job.close(); ctx.close()
        n = 10
        pipe = drv.DevicePipeline(device)
        for t in [pipe.submit(panos[i % 2], yaws, pitches, float(fov), ow, oh) for i in range(4)]:
            t.result()
        t0 = time.perf_counter()
        tickets = []
        for i in range(n):
            tickets.append(pipe.submit(panos[i % 2], yaws, pitches, float(fov), ow, oh))
            if len(tickets) > 2:
                tickets.pop(0).result()
        for t in tickets:
            t.result()
        ms = (time.perf_counter() - t0) / n * 1e3
        pipe.close()
        out["e2e_host_to_host"] = {
            "workload": w["name"] + ", page-locked host buffers in and out, two-slot device pipeline (no decode / encode)",
            "images": n, "ms_per_image": ms, "Gpix_s": len(yaws) * len(pitches) * ow * oh / ms / 1e6,
            "h2d_bytes": up_b, "d2h_bytes": down_b, "h2d_ms_alone": h2d_ms, "d2h_ms_alone": d2h_ms,
            "h2d_GBs_alone": up_b / h2d_ms / 1e6, "d2h_GBs_alone": down_b / d2h_ms / 1e6,
            "GBs_both_ways_in_pipeline": (up_b + down_b) / ms / 1e6,
            "bound": "the download of %.0f MB of uncompressed views (the kernel is %.1f %% of an image's time)" % (down_b / 1e6, 100 * 0.085 / ms)}
        nat.remap_views(panos[0], yaws, pitches, fov, ow, oh, device=device, pinned=True)
        ts = []
        for i in range(5):
            t0 = time.perf_counter()
            v = nat.remap_views(panos[i % 2], yaws, pitches, fov, ow, oh, device=device, pinned=True)
            ts.append(time.perf_counter() - t0)
            del v
        out["oneshot"] = {"workload": w["name"] + ", one p2p_remap_views_u8 call per image (upload, kernel, download in turn; geometry cached)",
                          "calls": len(ts), "ms_per_call_min": min(ts) * 1e3, "ms_per_call_median": sorted(ts)[len(ts) // 2] * 1e3,
                          "Gpix_s": len(yaws) * len(pitches) * ow * oh / (min(ts) * 1e3) / 1e6}
        nat.release_cache()
        U, V = pkg.get_pitch_mapping(ow, oh, 60, w["pw"], w["ph"], fov)
        pkg.panorama_to_plane(pano8k, U, V)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); v = pkg.panorama_to_plane(pano8k, U, V); ts.append(time.perf_counter() - t0); del v
        # the same remap's KERNEL alone (the table kernel: three channels, a non-constant border), apart from its uploads: the
        # legacy call as a resident job -- maps set once, the image resident (p2p_job_set_border + p2p_job_set_maps)
        k_ms = None
        try:
            ctx = nat.Context(device)
            rj = nat.Job(ctx, w["pw"], w["ph"], 1, [0.0], [90.0], 90.0, ow, oh)
            rj.set_border(nat.BORDER_REFLECT)
            rj.set_maps(None, U[None], V[None])
            rj.set_pano(0, panos[0])
            for _ in range(200):
                rj.run()
            rj.time_launches(64)
            for _ in range(64):
                rj.run()
            k_ms = float(np.median(rj.kernel_ms_last(64)))
            rj.close(); ctx.close()
        except Exception as e:
            k_ms = repr(e)
        out["legacy_reflect_3ch"] = {
            "workload": "legacy panorama_to_plane(pano, U, V) (L:159-194): one cv2.remap, BORDER_REFLECT, 8192x4096 -> one 1920x1080 view "
                        "(pitch 60), pageable host arrays in and out, maps uploaded per call",
            "calls": len(ts), "ms_per_call_min": min(ts) * 1e3, "ms_per_call_median": sorted(ts)[len(ts) // 2] * 1e3,
            "kernel_ms": k_ms, "upload_ms_of_the_panorama_alone": h2d_ms,
            "kernel_how": "the same remap as a resident job (p2p_job_set_border(REFLECT) + p2p_job_set_maps; what the legacy tool's "
                          "folder driver runs per image): median of 64 launches, HIP events",
            "Mpix_s": ow * oh / (min(ts) * 1e3) / 1e3,
            "bytes_per_call": {"source_up": up_b, "maps_up": 2 * 4 * ow * oh, "view_down": 3 * ow * oh},
            "bound": "the upload of the %.0f MB panorama on every call (the reference's signature hands it over each time)" % (up_b / 1e6)}
    except Exception as e:  # never takes the headline down
        out["host_to_host_error"] = repr(e)
    return out


def exact_route_line(pkg, nat, w, pano, device):
    """The identical-results route (--exact) on the metric's configuration: what its FIRST image pays on top of the default
    route -- the pitch maps evaluated on the host as the reference evaluates them (P:114-175), their upload, the plan made
    from them -- and what a launch costs in the steady state, where nothing but the kernels run (the maps stay with the
    job as they stay in the reference's pitch_mapping_cache, P:18)."""
    import numpy as np

    em = importlib.import_module(PKG + "._exact_maps")
    em.clear()
    t0 = time.perf_counter()
    U, V, key = em.pitch_map_stack(w["ow"], w["oh"], w["pitches"], w["pw"], w["ph"], w["fov"])
    host_maps_ms = (time.perf_counter() - t0) * 1e3
    ctx = nat.Context(device)
    try:
        job = nat.Job(ctx, w["pw"], w["ph"], 1, w["yaws"], w["pitches"], w["fov"], w["ow"], w["oh"])
        job.set_pano(0, pano)
        t0 = time.perf_counter()
        job.set_maps(None, U, V)
        upload_ms = (time.perf_counter() - t0) * 1e3
        job.time_launches(1)  # (plan_ms: the plan pass is timed for a job that asks)
        ctx.mark(0)
        job.run()
        ctx.mark(1)
        first_run_ms = ctx.marked_ms()
        plan_ms, tables_ms = job.plan_ms()
        job.time_launches(False)     # (the steady state below: no event inside a launch)
        t_pre = time.perf_counter()  # (the headline's own protocol: half a second of launches before the timed ones)
        while time.perf_counter() - t_pre < 0.5:
            for _ in range(50):
                job.run()
            ctx.synchronize()
        n = 1000
        ctx.mark(0)
        for _ in range(n):
            job.run()
        ctx.mark(1)
        steady_us = ctx.marked_ms() / n * 1e3
        info = job.info()
        job.close()
    finally:
        ctx.close()
        em.clear()
    b_alg = algorithmic_bytes(w, 1)
    return {"workload": w["name"] + ", --exact: pitch maps evaluated on the host (NumPy, the reference's float32 flow), pixels on the GPU",
            "host_maps_ms": host_maps_ms, "maps_upload_ms": upload_ms, "maps_bytes": int(U.nbytes + V.nbytes),
            "first_run_ms": first_run_ms, "plan_ms": plan_ms, "yaw_tables_ms": tables_ms,
            "first_image_ms": host_maps_ms + upload_ms + tables_ms + first_run_ms,
            "steady_state_us_per_launch": steady_us, "frac_of_hbm_peak": b_alg / (steady_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "n_gather_tiles": info["n_gather_tiles"],
            "how": "fresh context; host_maps_ms = wall time of _exact_maps.pitch_map_stack for the %d pitch maps; maps_upload_ms = "
                   "wall time of p2p_job_set_maps (pageable float32 arrays); first_run_ms / steady state = HIP events on the job's "
                   "stream; the reference pays its own map builders on its first image (0.27 s per pitch map at 1080p, SURVEY 3.5)"
                   % len(w["pitches"])}


def cold_first_image(nat, w, pano, device):
    """What ONE image through a context that has not seen its geometry pays on the device, next to the steady state
    the headline quotes: yaw tables + plan pass + view kernel (the reference's first image pays its map builders,
    P:79-175, the same way).  A fresh context, so that nothing comes from the table caches."""
    def one(timed):
        ctx = nat.Context(device)
        try:
            job = nat.Job(ctx, w["pw"], w["ph"], 1, w["yaws"], w["pitches"], w["fov"], w["ow"], w["oh"])
            job.set_pano(0, pano)
            if timed:
                job.time_launches(1)
            ctx.mark(0)
            job.run()   # plan pass (device), its gather-tile count handed to the host, the view kernel behind the pass
            ctx.mark(1)
            first = ctx.marked_ms()
            plan_ms, tables_ms = job.plan_ms()
            k = job.kernel_ms() if timed else None
            job.close()
            return first, plan_ms, tables_ms, k
        finally:
            ctx.close()

    first, _, tables_ms, _ = one(False)          # the product's path: no event inside the run
    first_t, plan_ms, tables_t, k = one(True)    # the same with the plan pass and the view kernel bracketed by events
    return {"plan_ms": plan_ms, "yaw_tables_ms": tables_ms, "first_run_ms": first, "view_kernel_ms_in_first_run": k,
            "cold_one_image_ms": tables_ms + first, "first_run_ms_with_timing_events": first_t,
            "how": "fresh context and job, twice: first_run_ms = HIP events around the first p2p_job_run of a job that records "
                   "no event of its own (the plan pass, the main kernel in grid order right behind it -- the block for the plan's "
                   "tables was fetched from the driver at job creation, while the device made the yaw tables; the pass's last workgroup hands the gather count to the host through page-locked memory, "
                   "so neither a copy nor that kernel is waited for; the per-XCD work lists are made when a second launch asks "
                   "for them); plan_ms / view_kernel_ms_in_first_run / first_run_ms_with_timing_events from a second fresh "
                   "context with p2p_job_time_launches on (four more events inside the run); yaw tables are built at job "
                   "creation (always timed)"}


def read_sclk_mhz():
    """Current shader clock of each GPU from sysfs (pp_dpm_sclk marks the active level with '*'), or None."""
    out = []
    for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            for line in open(path):
                if "*" in line:
                    out.append(int(line.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip()))
        except (OSError, ValueError, IndexError):
            pass
    return out or None


PMC_PASSES = (  # one rocprofv3 run each: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2
    ["FETCH_SIZE"], ["WRITE_SIZE"],
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"],
)


_PROFILER_ENV_PREFIXES = ("ROCP", "ROCPROF", "ROCTRACER", "ROCTX", "HSA_TOOLS", "ROCM_TOOLS")


def under_profiler():
    """True when THIS process was started by a profiler (rocprofv3 preloads its tool library and exports its own
    variables): then no counter children are spawned -- a rocprofv3 launcher inheriting that environment would
    initialise the GPU and exec its target, the hop this pool forbids."""
    env = os.environ
    if any("rocprof" in env.get(k, "").lower() or "roctracer" in env.get(k, "").lower()
           for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES")):
        return True
    return any(k.startswith(_PROFILER_ENV_PREFIXES) for k in env)


def clean_child_env(tmp):
    """Environment for the rocprofv3 children: this one minus anything a profiler may have put there."""
    env = {k: v for k, v in os.environ.items() if not k.startswith(_PROFILER_ENV_PREFIXES)}
    if "rocprof" in env.get("LD_PRELOAD", "").lower() or "roctracer" in env.get("LD_PRELOAD", "").lower():
        del env["LD_PRELOAD"]
    env["TMPDIR"] = tmp
    return env


def measure_counters(args):
    """HBM-side bytes and VALU instruction counts of the hot kernel of THIS build, per launch: this script re-run
    under `rocprofv3 --pmc` (counters only: no tracing in the same run), a few launches, one pass per counter
    group.  Returns {counter: mean per launch} or None when the profiler is not usable."""
    if under_profiler():
        return None
    kernel = "remap_views_kernel" if args.pixel_path == "u8" else "float_views_kernel"
    base = [sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "2", "--no-preroll", "--no-cpu-baseline", "--no-secondary",
            "--counters", "none", "--workload", args.workload, "--panos-per-gpu", str(args.panos_per_gpu),
            "--scaling", args.scaling,
            "--maps", args.maps, "--kind", args.kind, "--pixel-path", args.pixel_path]
    vals = {}
    tmp = tempfile.mkdtemp(prefix="p2p_pmc_")
    try:
        for i, group in enumerate(PMC_PASSES):
            d = os.path.join(tmp, "p%d" % i)
            try:
                subprocess.run(["rocprofv3", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--"] + base,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, cwd=tmp,
                               env=clean_child_env(tmp))
            except (OSError, subprocess.SubprocessError):
                continue
            acc, dur = {}, {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if kernel in row.get("Kernel_Name", ""):
                            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                            try:  # (one row per counter and dispatch: the dispatch's own duration, once)
                                dur[row.get("Dispatch_Id")] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                            except (KeyError, TypeError, ValueError):
                                pass
            for k, v in acc.items():
                vals[k] = sum(v) / len(v)
            good = [x for x in dur.values() if x > 0]
            if good and "SQ_BUSY_CU_CYCLES" in acc:  # the SQ pass: the clock this kernel held while it was counted
                vals["_sq_pass_kernel_ns"] = sum(good) / len(good)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return vals or None


def traffic_from_counters(c):
    """Guide (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory side; on gfx950
    FETCH_SIZE tallies the 128-B requests of wide reads at 64 B.  The factor for this kernel's read pattern
    (4-byte-aligned 16-byte pieces 12 bytes apart) was calibrated on a known-bytes read, tools/ubench/fetch_calib.hip:
    x1.92 (x2.00 for plain dwordx4 streaming); WRITE_SIZE is used as reported."""
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    corr = 1.92
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            corr = float(json.load(f).get("cfg2", {}).get("fetch_correction", corr))
    except (OSError, ValueError):
        pass
    rd, wr = c["FETCH_SIZE"] * 1024.0 * corr, c["WRITE_SIZE"] * 1024.0
    return {"bytes": rd + wr, "read_bytes": rd, "write_bytes": wr, "fetch_correction": corr}


def load_traffic(workload):
    """HBM bytes per launch from the PMC passes (profiles/traffic.json, written from rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE runs with the guide's gfx950 corrections), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed launches (default 2000 = 0.3 s of GPU time on config 2: long enough for the clock to settle)")
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: cfg2 at --gpus 1 (the configuration the metric is quoted on); cfg3 at --gpus N > 1 "
                         "(the 64-panorama batch, 64 / N panoramas resident per GPU: SURVEY 8(e) reads scaling there)")
    ap.add_argument("--panos-per-gpu", type=int, default=None,
                    help="panoramas resident per GPU (default 1; cfg3: its share of the 64)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: every rank gets --panos-per-gpu panoramas of its own (default for cfg2 / cfg4 / cfg5); "
                         "strong: the workload's total is split -- cfg3's 64 panoramas dealt to the ranks (its default), "
                         "or, for a single-panorama workload, its (yaw x pitch) views in pitch-major runs")
    ap.add_argument("--maps", default="fused", choices=["fused", "caller"],
                    help="fused: coordinate maps computed in-kernel (default, the product path); "
                         "caller: float maps handed in (the bit-exact mode)")
    ap.add_argument("--kind", default="S", choices=["S", "N"], help="synthetic panorama distribution")
    ap.add_argument("--pixel-path", default="u8", choices=["u8", "f32", "f16"],
                    help="u8: the reference's two fixed-point remap stages (default; the parity path). "
                         "f32 / f16: the opt-in single float resample, which the reference does not have")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary lines (configs 3-share / 4 / 5, the reference CLI's default view set) and the "
                         "cold-image figures the default cfg2 run at --gpus 1 appends")
    ap.add_argument("--preroll-s", type=float, default=0.5,
                    help="seconds of untimed launches BEFORE the --warmup steps, whatever --steps / --warmup are: an idle "
                         "MI355X sits at a low shader clock and needs a few hundred ms of load to reach its working "
                         "clock (a 25-launch run is over in 4 ms).  Reported as preroll_s; 0 turns it off")
    ap.add_argument("--no-preroll", action="store_true")
    ap.add_argument("--shard", default="rows", choices=["rows", "views"],
                    help="--scaling strong on one panorama: a band of rows of every view per rank (default) or whole views per rank")
    ap.add_argument("--counters", default="auto", choices=["auto", "measure", "file", "none"],
                    help="roofline.traffic / roofline.valu: measure = rocprofv3 --pmc child runs of this script (one pass "
                         "per counter group, before this process touches the GPU); file = profiles/traffic.json; "
                         "auto = measure at --gpus 1 when rocprofv3 is on PATH and this process is not itself running "
                         "under a profiler, else file")
    ap.add_argument("--host-to-host-only", action="store_true",
                    help="print the host-buffer secondary lines (e2e_host_to_host, oneshot, legacy_reflect_3ch) as one JSON "
                         "object and exit: what the default run starts as a child process")
    args = ap.parse_args()
    if args.host_to_host_only:
        pkg = importlib.import_module(PKG)
        synth = importlib.import_module(PKG + ".synth")
        drv = importlib.import_module(PKG + "._driver")
        print(json.dumps(host_to_host_lines(pkg, pkg._native, drv, synth.synth_pano(8192, 4096, 1000, args.kind), 0)))
        return
    if args.no_preroll:
        args.preroll_s = 0.0
    if args.workload is None:
        # the metric's configuration at every N: images dealt to the ranks, one resident panorama x 36 views each
        # (weak scaling: the same per-GPU workload as the N = 1 line; config 3's 64-panorama batch is --workload cfg3)
        args.workload = "cfg2"
    if args.scaling is None:
        args.scaling = "strong" if "n_panos_total" in WORKLOADS[args.workload] else "weak"
    if args.panos_per_gpu is None:
        args.panos_per_gpu = 1
    counters = None
    mode = args.counters
    if mode == "auto":
        mode = "measure" if (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and shutil.which("rocprofv3")
                             and not under_profiler()) else "file"
    if mode == "measure":
        counters = measure_counters(args)  # child processes; nothing in THIS process has touched the GPU yet
    h2h = None
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.workload == "cfg2" and not args.no_secondary and \
            args.pixel_path == "u8" and args.maps == "fused" and args.scaling == "weak" and args.panos_per_gpu == 1 and not under_profiler():
        try:  # the host-to-host secondary lines, in a fresh process (see below)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-to-host-only", "--kind", args.kind],
                               capture_output=True, text=True, timeout=180, env=clean_child_env(tempfile.gettempdir()))
            h2h = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            h2h = None

    import torch

    dist = Dist()
    if dist.world != args.gpus:
        if dist.rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, dist.world),
                  file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU fallback for the hot path)"
    if os.environ.get("P2P_BENCH_SHARE_GPU") == "1":  # dry run only: several ranks on one device
        dist.local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(dist.local_rank)

    pkg = importlib.import_module(PKG)
    nat = pkg._native
    synth = importlib.import_module(PKG + ".synth")
    drv = importlib.import_module(PKG + "._driver")
    w = WORKLOADS[args.workload]
    flags = {"u8": 0, "f32": nat.FLAG_PIXELS_F32, "f16": nat.FLAG_PIXELS_F16}[args.pixel_path]
    n_yaw, n_pitch = len(w["yaws"]), len(w["pitches"])
    ctx = nat.Context(dist.local_rank)
    jobs = []          # what one step launches on this rank
    sharding = None
    if args.scaling == "strong" and "n_panos_total" in w:
        # config 3: the batch's panoramas dealt round-robin to the ranks, each rank keeps its share resident
        mine = shard_round_robin(w["n_panos_total"], dist.world, dist.rank)
        npg, total_views = len(mine), w["n_panos_total"] * n_yaw * n_pitch
        seeds = [1000 + i for i in mine]
        sharding = "%d panoramas dealt round-robin, %d resident per GPU, no collective" % (w["n_panos_total"], npg)
        if npg:
            jobs.append(nat.Job(ctx, w["pw"], w["ph"], npg, w["yaws"], w["pitches"], w["fov"], w["ow"], w["oh"], flags=flags))
        views_per_rank = npg * n_yaw * n_pitch
    elif args.scaling == "strong" and args.shard == "rows":
        # one panorama, every rank draws a band of ROWS of every view (drv.shard_rows, p2p_job_set_rows): a tile's set-up
        # stays spread over all the yaws, and the number of views does not cap the speed-up; every rank uploads the panorama
        r0, r1 = drv.shard_rows(w["oh"], dist.world, w["pitches"], w["fov"], w["ow"])[dist.rank]
        npg, total_views, seeds = 1, n_yaw * n_pitch, [1000]
        if r1 > r0:
            j = nat.Job(ctx, w["pw"], w["ph"], 1, w["yaws"], w["pitches"], w["fov"], w["ow"], w["oh"], flags=flags)
            j.set_rows(r0, r1)
            jobs.append(j)
        views_per_rank = n_yaw * n_pitch * (r1 - r0) / float(w["oh"])  # (in whole views' worth of pixels)
        sharding = "rows %d..%d of every view of one panorama on this rank, no collective" % (r0, r1)
    elif args.scaling == "strong":
        # one panorama, its pitch-major view list cut into one contiguous run per rank (36 views on 8 GPUs
        # = 5 or 4 consecutive yaws of one pitch view each); every rank uploads the panorama once
        yaw_idx, pitch_idx, mask, mine_views = drv.rank_view_set(n_yaw, n_pitch, dist.world, dist.rank, pitch_deg=w["pitches"])
        npg, total_views, seeds = 1, n_yaw * n_pitch, [1000]
        if mine_views:
            # ONE job per rank: the yaws and pitches that occur in its share, and a view mask for the combinations that
            # are its own (p2p_job_set_view_mask) -- one launch whose pitch views share the source rows they read
            j = nat.Job(ctx, w["pw"], w["ph"], 1, [w["yaws"][y] for y in yaw_idx], [w["pitches"][p] for p in pitch_idx],
                        w["fov"], w["ow"], w["oh"], flags=flags)
            if not mask.all():
                j.set_view_mask(mask)
            jobs.append(j)
        views_per_rank = len(mine_views)
        sharding = "views of one panorama, pitch-major runs, %d on this rank in one job, no collective" % views_per_rank
    else:
        npg = args.panos_per_gpu
        total_views = npg * n_yaw * n_pitch * dist.world
        # weak scaling: panorama index = rank * panos_per_gpu + i, seed 1000 + index
        seeds = [1000 + dist.rank * npg + i for i in range(npg)]
        jobs.append(nat.Job(ctx, w["pw"], w["ph"], npg, w["yaws"], w["pitches"], w["fov"], w["ow"], w["oh"], flags=flags))
        views_per_rank = npg * n_yaw * n_pitch
        sharding = "independent panoramas per rank, no collective"
    job = jobs[0] if jobs else None
    if jobs:
        # synthetic panoramas, seed 1000 + index; generated on a few host threads (NumPy releases the
        # GIL): config 3's share is up to 64 of them
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=max(1, min(8, len(seeds), (os.cpu_count() or 1) // max(1, dist.world)))) as ex:
            for i, pano in enumerate(ex.map(lambda sd: synth.synth_pano(w["pw"], w["ph"], sd, args.kind), seeds)):
                jobs[0].set_pano(i, pano)
    if args.maps == "caller":
        import numpy as np  # float maps from the library's own device map builders, handed back in

        assert len(jobs) == 1 and args.scaling != "strong", "--maps caller is a single-job mode"
        rows = np.stack([nat.build_yaw_row(w["pw"], float(np.radians(y)), dist.local_rank) for y in w["yaws"]])
        UV = [nat.build_pitch_map(w["ow"], w["oh"], float(np.radians(w["fov"])), float(np.radians(p)),
                                  w["pw"], w["ph"], dist.local_rank) for p in w["pitches"]]
        job.set_maps(rows, np.stack([u for u, _ in UV]), np.stack([v for _, v in UV]))

    def device_sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    # The timed region is bracketed by two HIP events on the job's stream (ctx.mark), so the K launches run
    # back to back with nothing between them; the kernel's mean duration = marked time / K (it includes the
    # ~2 us launch-to-launch boundary).  Per-launch event pairs -- two more event records per launch -- are
    # switched off inside the region and used afterwards on a short sample to report the spread.
    steps = args.steps
    for j in jobs:
        j.time_launches(False)
    state = {"n": 0}

    def step():
        if state["n"] == args.warmup:
            ctx.mark(0)
        for j in jobs:
            j.run()
        state["n"] += 1
        if state["n"] == args.warmup + steps:
            ctx.mark(1)

    # disclosed pre-roll: launches for at least --preroll-s seconds before the --warmup steps, so that the timed
    # region sees the clock the GPU holds under this load and not the ramp from idle
    sclk_before = read_sclk_mhz()
    preroll_launches, t_pre = 0, time.perf_counter()
    while time.perf_counter() - t_pre < args.preroll_s:
        for _ in range(50):
            for j in jobs:
                j.run()
        ctx.synchronize()
        preroll_launches += 50
    preroll_s = time.perf_counter() - t_pre if preroll_launches else 0.0
    elapsed = run_timed(step, device_sync, dist, steps, args.warmup)
    sclk_after = read_sclk_mhz()
    k_avg_s = ctx.marked_ms() / steps / 1e3  # all of this rank's launches of one step
    import numpy as np
    sample = min(32, steps)
    kms = np.zeros(sample, np.float32)
    for j in jobs:
        j.time_launches(True)
        for _ in range(sample):
            j.run()
        kms = kms + j.kernel_ms_last(sample)

    # what a plain device-to-device copy reaches on this GPU right now (read + written bytes per second):
    # the practical ceiling next to the 8 TB/s specification peak
    copy_gbs = None
    try:
        a = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            b.copy_(a)
        ev0.record()
        for _ in range(10):
            b.copy_(a)
        ev1.record()
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * a.numel() / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
        del a, b
    except Exception:
        pass

    pix_per_step = total_views * w["ow"] * w["oh"]
    value = pix_per_step * steps / elapsed / 1e6
    # this rank's share: its resident panoramas read once, its views written once
    b_alg = 3 * w["pw"] * w["ph"] * (npg if jobs else 0) + 3 * w["ow"] * w["oh"] * views_per_rank
    achieved = b_alg / k_avg_s / 1e9
    if mode == "measure":
        tr = traffic_from_counters(counters)
        traffic = tr["bytes"] if tr else None
        traffic_how = dict(tr, source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this build") if tr else \
            {"source": "rocprofv3 passes failed"}
    elif mode == "file":
        traffic, traffic_how = load_traffic(args.workload), {"source": "profiles/traffic.json (an earlier build's passes)"}
    else:
        traffic, traffic_how = None, {"source": "not collected"}
    valu = None
    if counters and "SQ_INSTS_VALU" in counters:
        px = views_per_rank * w["ow"] * w["oh"]
        # SQ_WAVES counts every wave of the launch (config 2: 24 480): the figures below are per launch, unscaled
        waves = counters.get("SQ_WAVES")
        busy = counters.get("SQ_BUSY_CU_CYCLES")
        valu = {"SQ_INSTS_VALU": counters["SQ_INSTS_VALU"], "SQ_INSTS_SALU": counters.get("SQ_INSTS_SALU"),
                "SQ_WAVES": waves,
                "valu_wave_insts_per_wave": counters["SQ_INSTS_VALU"] / waves if waves else None,
                "valu_lane_insts_per_output_px": (counters["SQ_INSTS_VALU"] * 64.0 / px) if px else None,
                "active_inst_valu_over_busy_cu_cycles": (counters.get("SQ_ACTIVE_INST_VALU", 0.0) / busy) if busy else None,
                "issue_ceiling_of_that_ratio": {"slow_class_only": 0.97, "fast_class_only": 1.80,
                                                "source": "profiles/r01_valu_counter_calibration.txt"},
                "note": "secondary ceilings: the exact two-stage fixed-point emulation keeps the VALU issue ports about "
                        "0.8 busy and moves traffic_rate_GBs through the L2 fabric at the same time (a plain copy "
                        "reaches measured_copy_GBs): co-limited, DESIGN.md 5.2"}
    # The kernel's own issue floor: its vector instructions, all issued at one wave-instruction per
    # 4 cycles and SIMD, on the 1024 SIMDs, at the clock the kernel itself holds -- SQ_BUSY_CU_CYCLES (summed over the 256
    # CUs) over the duration of the counted dispatches; without those timestamps, over this run's mean launch duration.
    issue = None
    if counters and counters.get("SQ_INSTS_VALU") and counters.get("SQ_BUSY_CU_CYCLES"):
        n_cu, n_simd = 256, 1024
        dur_s = counters["_sq_pass_kernel_ns"] * 1e-9 if counters.get("_sq_pass_kernel_ns") else k_avg_s
        clock_hz = counters["SQ_BUSY_CU_CYCLES"] / n_cu / dur_s
        if clock_hz > 0:
            floor_us = counters["SQ_INSTS_VALU"] * 4.0 / n_simd / clock_hz * 1e6
            issue = {"issue_floor_us": floor_us, "frac_of_issue_floor": floor_us / (k_avg_s * 1e6),
                     "clock_ghz_from_busy_cu_cycles": clock_hz / 1e9,
                     "clock_from": "SQ_BUSY_CU_CYCLES / 256 CUs / " + ("the counted dispatches' own duration (%.1f us under the profiler)"
                                   % (dur_s * 1e6) if counters.get("_sq_pass_kernel_ns") else "this run's mean launch duration"),
                     "how": "SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / that clock; frac_of_issue_floor = issue_floor_us / kernel_ms_avg: "
                            "how much of a launch is vector-instruction issue at a perfect schedule",
                     "target_0p70_us": b_alg / (0.70 * HBM_PEAK_GBS * 1e9) * 1e6}
    out = {
        "metric": "Mpix/s remapped, 8K equirect->1080p x36 views" if args.workload == "cfg2"
                  else "Mpix/s remapped (%s)" % args.workload,
        "value": value, "unit": "Mpix/s", "n_gpus": dist.world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": args.pixel_path, "data": "synthetic",
        "config": {"workload": w["name"], "panos_per_gpu": npg, "views_per_gpu": views_per_rank,
                   "maps": {"fused": "device plan (evaluated once per job geometry, like the reference's pitch_mapping_cache)",
                            "caller": "caller float maps (plan built from them once)"}[args.maps], "pano_kind": args.kind, "sharding": sharding, "launches_per_step": len(jobs)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_how": traffic_how, "valu": valu,
                     "kernel": "remap_views_kernel" if args.pixel_path == "u8" else "float_views_kernel", "kernel_ms_avg": k_avg_s * 1e3,
                     "kernel_ms_sample": {"n": int(sample), "mean": float(kms.mean()), "min": float(kms.min()),
                                          "max": float(kms.max()), "how": "own HIP event pair per launch, after the timed region"},
                     "algorithmic_bytes_per_launch": b_alg,
                     "traffic_rate_GBs": (traffic / k_avg_s / 1e9) if (traffic and k_avg_s) else None,
                     "measured_copy_GBs": copy_gbs,
                     "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None,
                     "issue_floor_us": issue["issue_floor_us"] if issue else None,
                     "frac_of_issue_floor": issue["frac_of_issue_floor"] if issue else None,
                     "issue_floor": issue},
        "preroll_s": preroll_s, "preroll_launches": preroll_launches,
        "sclk_mhz": {"before_preroll": sclk_before, "after_timed_region": sclk_after},
    }
    if dist.rank == 0 and dist.world == 1 and args.workload == "cfg2" and args.scaling == "weak" and npg == 1 and \
            args.pixel_path == "u8" and args.maps == "fused" and not args.no_secondary:
        pano8k = synth.synth_pano(8192, 4096, 1000, args.kind)
        out["cold"] = cold_first_image(nat, w, pano8k, dist.local_rank)
        out["secondary"] = secondary_lines(nat, ctx, pano8k, dist.local_rank)
        try:
            out["secondary"]["exact_route"] = exact_route_line(pkg, nat, w, pano8k, dist.local_rank)
        except Exception as e:  # a secondary line never takes the headline down
            out["secondary"]["exact_route"] = {"error": repr(e)}
        # the host-buffer lines come from a process of their own, run before this one touched the GPU (h2h below): in
        # THIS process -- torch initialised, a dozen streams created and destroyed by now -- the pipeline's three
        # streams no longer overlap (9.3 ms per image against 4.5: the runtime maps streams onto a handful of
        # hardware queues); in-process only when no child could be started
        out["secondary"].update(h2h if h2h else host_to_host_lines(pkg, nat, drv, pano8k, dist.local_rank))
        del pano8k
    if dist.world > 1 and args.workload == "cfg2" and args.scaling == "weak" and not args.no_secondary and args.pixel_path == "u8":
        # One number about the SHARDING code next to the weak-scaling headline (which is N x the one-GPU workload by
        # construction): BASELINE config 3, the 64-panorama batch dealt round-robin, 64 / N resident per GPU, a few
        # launches bracketed like the headline (barrier, device sync, max over ranks).  Every rank takes part.
        w3 = WORKLOADS["cfg3"]
        mine3 = shard_round_robin(w3["n_panos_total"], dist.world, dist.rank)
        sec = {"workload": w3["name"], "scaling": "strong", "panos_per_gpu": len(mine3)}
        job3, failed = None, 0.0
        try:
            if mine3:
                job3 = nat.Job(ctx, w3["pw"], w3["ph"], len(mine3), w3["yaws"], w3["pitches"], w3["fov"], w3["ow"], w3["oh"])
                pano3 = synth.synth_pano(w3["pw"], w3["ph"], 1000 + dist.rank, args.kind)  # (content does not enter the timing)
                for i in range(len(mine3)):
                    job3.set_pano(i, pano3)
                job3.time_launches(False)
                job3.run()
                ctx.synchronize()
        except Exception as e:
            failed, sec["error"] = 1.0, repr(e)
        # every rank learns whether ALL ranks are ready before anyone enters the barriers of the timed loop: a rank that
        # failed above must not leave the others waiting in one
        if dist.max_over_ranks(failed) == 0.0:
            k3 = 6
            el3 = run_timed((lambda: job3.run()) if job3 is not None else (lambda: None), device_sync, dist, k3, 2)
            views3 = w3["n_panos_total"] * len(w3["yaws"]) * len(w3["pitches"])
            sec.update({"steps": k3, "ms_per_step": el3 / k3 * 1e3, "value_Mpix_s": views3 * w3["ow"] * w3["oh"] * k3 / el3 / 1e6,
                        "algorithmic_bytes_per_gpu": algorithmic_bytes(w3, len(mine3)),
                        "frac_of_hbm_peak_per_gpu": algorithmic_bytes(w3, len(mine3)) / (el3 / k3) / 1e9 / HBM_PEAK_GBS})
        else:
            sec.setdefault("error", "another rank could not set its share up")
        if job3 is not None:
            try:
                job3.close()
            except Exception:
                pass
        out.setdefault("secondary", {})["cfg3_strong"] = sec
    if dist.rank == 0 and dist.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w)
    elif dist.rank == 0:
        out["cpu_baseline"] = None
    for j in reversed(jobs):
        j.close()
    ctx.close()
    dist.close()
    if dist.rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
