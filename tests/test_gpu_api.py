"""GPU: the drop-in Python surface end to end (shapes, ordering, thread-pool re-entrancy, error codes, files)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from _util import diff_stats, oracle_views

pytestmark = pytest.mark.gpu


def test_process_yaw_and_pitchs_contract(gpu, pkg, synth):
    pano = synth.synth_pano(1024, 512, 3000, "S")
    out = pkg.process_yaw_and_pitchs(pano, 30, [60, 90, 120], 160, 120, 90)
    assert isinstance(out, list) and len(out) == 3
    for o in out:
        assert o.shape == (120, 160, 3) and o.dtype == np.uint8
    want = oracle_views(pano, [30], [60, 90, 120], 160, 120, 90)
    assert diff_stats(np.stack(out)[None], want)[0] <= 1
    # default FOV 90, numpy integer angles, float angles that are whole numbers
    a = pkg.process_yaw_and_pitchs(pano, np.int64(30), [np.int32(90)], 160, 120)
    b = pkg.process_yaw_and_pitchs(pano, 30.0, [90], 160, 120, fov_deg=90)
    assert np.array_equal(a[0], out[1]) and np.array_equal(b[0], out[1])
    with pytest.raises(TypeError):
        pkg.process_yaw_and_pitchs(pano, "30", [90], 160, 120)


def test_real_valued_angles_like_the_reference_functions(gpu, pkg, synth):
    """P:85 and P:64-68 hand yaw, pitch and FOV to np.radians: any real number is legal input to the reference's
    functions (only its CLI narrows them).  Caller-map mode on the oracle's maps is bit-exact, fused mode within +-1."""
    from _util import oracle_maps
    pw, ph, ow, oh = 1024, 512, 200, 120
    yaws, pitches, fov = [30.5, -0.25, 359.9], [44.5, 90.0, 0.5, 179.75], 72.5
    for kind in ("N", "S"):
        pano = synth.synth_pano(pw, ph, 3050, kind)
        want = oracle_views(pano, yaws, pitches, ow, oh, fov)
        rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
        assert np.array_equal(gpu.remap_views_maps(pano, rows, U, V), want)
        if kind == "S":
            got = pkg.process_views(pano, yaws, pitches, ow, oh, fov)
            assert diff_stats(got, want)[0] <= 1
    one = pkg.process_yaw_and_pitchs(pano, 30.5, [44.5], ow, oh, fov_deg=72.5)
    assert np.array_equal(one[0], got[0, 0])
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
    job.set_pano(0, pano)
    job.run()
    assert np.array_equal(job.get_views(0), got)
    job.set_yaws([10.125, 20.25, 30.5])
    job.run()
    assert np.array_equal(job.get_views(0)[2], got[0])
    job.close()
    ctx.close()


def test_more_than_64_pitch_angles_in_one_call(gpu, pkg, synth):
    # P:424-430: --pitch_angles takes any list of 1..179; `seq 1 2 179` is 90 of them
    pano = synth.synth_pano(512, 256, 3060, "S")
    pitches = list(range(1, 180, 2)) + [90, 45, 17, 163, 2, 4, 6, 8, 10, 12]
    assert len(pitches) == 100
    got = pkg.process_views(pano, [0, 77], pitches, 64, 48, 90)
    assert got.shape == (2, 100, 48, 64, 3)
    for pi in (0, 1, 44, 45, 89, 90, 99):
        want = oracle_views(pano, [0, 77], [pitches[pi]], 64, 48, 90)
        assert diff_stats(got[:, pi:pi + 1], want)[0] <= 1, pitches[pi]


def test_thread_pool_fan_out_like_the_reference(gpu, pkg, synth):
    # P:252-265: one task per yaw on a shared panorama, up to int(0.9 * cores) threads
    pano = synth.synth_pano(1024, 512, 3001, "N")
    yaws, pitches = list(range(0, 360, 30)), [60, 120]
    serial = [pkg.process_yaw_and_pitchs(pano, y, pitches, 96, 64) for y in yaws]
    with ThreadPoolExecutor(max_workers=8) as ex:
        futs = [ex.submit(pkg.process_yaw_and_pitchs, pano, y, pitches, 96, 64) for y in yaws]
        par = [f.result() for f in futs]
    for s, p in zip(serial, par):
        for a, b in zip(s, p):
            assert np.array_equal(a, b)
    batched = pkg.process_views(pano, yaws, pitches, 96, 64)
    for yi in range(len(yaws)):
        for pi in range(len(pitches)):
            assert np.array_equal(batched[yi, pi], serial[yi][pi])


def test_error_codes(gpu, pkg, synth):
    pano = synth.synth_pano(64, 32, 3002, "N")
    # the integer entry points keep the CLI's pitch check (check_pitch, P:362-376); the functions themselves
    # take any pitch, as the reference's do
    for bad in (0, 180, -5):
        with pytest.raises(gpu.P2PError) as e:
            gpu.remap_views(pano, [0], [bad], 90, 16, 16)
        assert e.value.code == gpu.P2P_ERR_INVALID and "between 1 and 179" in str(e.value)
        with pytest.raises(gpu.P2PError) as e:
            gpu.Job(gpu.Context(0), 64, 32, 1, [0], [bad], 90, 16, 16, integer_abi=True)
        assert e.value.code == gpu.P2P_ERR_INVALID
    assert pkg.process_yaw_and_pitchs(pano, 0, [0, 180], 16, 16)[0].shape == (16, 16, 3)
    with pytest.raises(ValueError):
        pkg.process_yaw_and_pitchs(pano, float("nan"), [90], 16, 16)
    with pytest.raises(gpu.P2PError) as e:
        gpu.remap_views(pano, [0], [90], 90, 40000, 16)
    assert e.value.code == gpu.P2P_ERR_INVALID
    with pytest.raises(gpu.P2PError) as e:
        gpu.remap_views(pano, [0], [90], 90, 16, 16, device=99)
    assert e.value.code == gpu.P2P_ERR_NO_DEVICE
    with pytest.raises(gpu.P2PError) as e:
        gpu.remap_views_maps(pano, np.full((1, 64), 64.0, np.float32), np.zeros((1, 4, 4), np.float32),
                             np.zeros((1, 4, 4), np.float32))
    assert e.value.code == gpu.P2P_ERR_INVALID  # yaw row outside [0, pw-1]
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, 64, 32, 2, [0], [90], 90, 16, 16)
    job.set_pano(0, pano)
    with pytest.raises(gpu.P2PError) as e:
        job.run()  # panorama 1 was never set
    assert e.value.code == gpu.P2P_ERR_STATE
    assert pkg.process_views(pano, [], [90], 16, 16).shape == (0, 1, 16, 16, 3)
    job.close()
    ctx.close()


def test_cli_end_to_end_files(gpu, pkg, synth, tmp_path):
    from PIL import Image

    m = pkg.panorama_to_plane_pitch
    pano = synth.synth_pano(512, 256, 3003, "S")  # BGR in memory
    (tmp_path / "in").mkdir()
    Image.fromarray(pano[:, :, ::-1]).save(tmp_path / "in" / "room.png")
    m.cli(["--input_path", str(tmp_path / "in"), "--output_path", str(tmp_path / "out"), "--output_width", "64",
           "--output_height", "48", "--yaw_angles", "0", "77", "--pitch_angles", "60", "120", "--FOV", "100"])
    want = oracle_views(pano, [0, 77], [60, 120], 64, 48, 100)
    for yi, y in enumerate((0, 77)):
        for pi, p in enumerate((60, 120)):
            f = tmp_path / "out" / f"room_64x48_yaw_{y}_pitch_{p}.png"
            assert f.exists(), f
            got = np.asarray(Image.open(f).convert("RGB"))[:, :, ::-1]
            assert diff_stats(got, want[yi, pi])[0] <= 1


def test_maps_cache_keys_and_shapes(gpu, pkg):
    m = pkg.panorama_to_plane_pitch
    m.pitch_mapping_cache.clear()
    m.yaw_mapping_cache.clear()
    U, V = m.get_pitch_mapping(64, 48, 60, 512, 256, 90)
    assert (64, 48, 60, 512, 256, 90) in m.pitch_mapping_cache and U.shape == (48, 64)
    assert m.get_pitch_mapping(64, 48, 60, 512, 256)[0] is U
    Uy, Vy = m.get_yaw_mapping(512, 16, 30)
    assert (512, 16, 30) in m.yaw_mapping_cache and Uy.shape == (16, 512) and Vy[5, 7] == 5.0


def test_oneshot_cache_reuses_buffers_without_changing_results(gpu, pkg, synth, monkeypatch, p2p_env):
    """The one-shot calls keep the last geometry's device buffers per thread (the reference's map caches,
    P:17-18); every sequence of calls must give what a cold library gives."""
    from _util import oracle_maps

    pa = synth.synth_pano(512, 256, 3100, "N")
    pb = synth.synth_pano(512, 256, 3101, "N")
    pitches = [60, 90]

    def cold(pano, yaws, pit=pitches, ow=96, oh=64):
        p2p_env("P2P_ONESHOT_CACHE", "0")
        gpu.release_cache()
        try:
            return gpu.remap_views(pano, yaws, pit, 90, ow, oh)
        finally:
            p2p_env("P2P_ONESHOT_CACHE", "1")

    ref_a = cold(pa, [0, 77])
    ref_b = cold(pb, [13, 200])
    ref_small = cold(pa, [0, 77], ow=48, oh=32)
    gpu.release_cache()
    assert np.array_equal(gpu.remap_views(pa, [0, 77], pitches, 90, 96, 64), ref_a)           # miss
    assert np.array_equal(gpu.remap_views(pa, [0, 77], pitches, 90, 96, 64), ref_a)           # hit, same yaws
    assert np.array_equal(gpu.remap_views(pb, [13, 200], pitches, 90, 96, 64), ref_b)         # hit, new yaws + pano
    assert np.array_equal(gpu.remap_views(pa, [0, 77], pitches, 90, 48, 32), ref_small)       # other geometry
    assert np.array_equal(gpu.remap_views(pa, [0, 77], pitches, 90, 96, 64, pinned=True), ref_a)
    # caller maps <-> in-kernel maps alternate on one geometry; caller rows must not leak into the next call
    rows, U, V = oracle_maps([5, 300], pitches, 96, 64, 512, 256)
    m1 = gpu.remap_views_maps(pa, rows, U, V)
    assert np.array_equal(gpu.remap_views(pa, [0, 77], pitches, 90, 96, 64), ref_a)
    m2 = gpu.remap_views_maps(pa, rows, U, V)
    m3 = gpu.remap_views_maps(pb, rows[::-1].copy(), U, V)
    assert np.array_equal(m1, m2)
    assert np.array_equal(m3[::-1], gpu.remap_views_maps(pb, rows, U, V))
    # different pitch list of the same length is a different geometry
    assert np.array_equal(gpu.remap_views(pa, [0, 77], [61, 90], 90, 96, 64), cold(pa, [0, 77], pit=[61, 90]))
    gpu.release_cache()


def test_job_set_yaws(gpu, synth):
    pano = synth.synth_pano(1024, 512, 3102, "N")
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, 1024, 512, 1, [0, 30, 45], [70, 110], 90, 128, 96)
    job.set_pano(0, pano)
    job.run()
    first = job.get_views()
    job.set_yaws([181, 30, 359])
    job.run()
    second = job.get_views(pinned=True)
    fresh = gpu.remap_views(pano, [181, 30, 359], [70, 110], 90, 128, 96)
    assert np.array_equal(second, fresh)
    assert np.array_equal(second[1], first[1])
    with pytest.raises(ValueError):
        job.set_yaws([1, 2])
    job.close()
    ctx.close()


def test_pinned_arrays_behave_like_numpy_arrays(gpu, pkg, synth):
    a = gpu.pinned_empty((5, 7, 3))
    assert a.shape == (5, 7, 3) and a.dtype == np.uint8 and a.flags.c_contiguous and a.flags.writeable
    a[...] = 7
    b = a[1:3]
    del a                       # the block must stay alive while a view of it is
    assert int(b.sum()) == 7 * 2 * 7 * 3
    f = gpu.pinned_empty((4,), np.float32)
    f[:] = 1.5
    assert f.sum() == 6.0
    # the drop-in API hands out page-locked views by default; results equal the pageable path
    pano = synth.synth_pano(512, 256, 3103, "N")
    pin = gpu.pinned_empty(pano.shape)
    pin[...] = pano
    v1 = pkg.process_views(pin, [10, 20], [80], 64, 48)
    v2 = gpu.remap_views(pano, [10, 20], [80], 90, 64, 48, pinned=False)
    assert np.array_equal(v1, v2)
    lst = pkg.process_yaw_and_pitchs(pano, 10, [80], 64, 48)
    del v1
    assert np.array_equal(lst[0], v2[0, 0])


def test_pinned_budget_falls_back_to_ordinary_arrays(gpu, synth, monkeypatch):
    pano = synth.synth_pano(256, 128, 3104, "N")
    want = gpu.remap_views(pano, [0, 90], [90], 90, 64, 48)
    monkeypatch.setattr(gpu._pool, "max_live", gpu._pool.live_bytes + 100)   # less than one result
    got = gpu.remap_views(pano, [0, 90], [90], 90, 64, 48, pinned=True)
    assert np.array_equal(got, want)
    with pytest.raises(MemoryError):
        gpu.pinned_empty((1 << 20,))


def test_view_sharded_path_two_contexts_on_one_device(gpu, pkg, synth):
    """SURVEY 8(e) / P:252-265: with fewer images than GPUs the views of an image are dealt to the devices.  Two
    contexts on device 0 stand in for two GPUs: the stitched result equals the single-device one byte for byte."""
    import importlib
    d = importlib.import_module("360-to-planer-images_amd._driver")
    pano = synth.synth_pano(2048, 1024, 3100, "N")
    yaws, pitches = list(range(0, 360, 30)), [60, 90, 120]
    one = pkg.process_views(pano, yaws, pitches, 320, 180, 90)
    for how in ("rows", "views", "auto"):  # a band of rows of every view per device | whole views per device
        for devices in ([0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [0]):
            got = d.process_views_sharded(pano, yaws, pitches, 320, 180, 90.0, devices, how=how)
            assert np.array_equal(got, one), (how, devices)
    # 36 views on 8 ranks: each rank's 4 or 5 views are ONE job with a view mask (a 3 x 3 grid with holes); and a view
    # width that is not divisible by 4 (single views through the job's packing buffer)
    one_odd = pkg.process_views(pano, yaws, pitches, 318, 180, 90)
    for how in ("views", "rows"):
        got = d.process_views_sharded(pano, yaws, pitches, 318, 180, 90.0, [0] * 8, how=how)
        assert np.array_equal(got, one_odd), how
    # a caller whose device list changes length from image to image: the contexts kept are those of the LAST call's
    # slots (device, k) -- never one per (rank, device) pair ever used (round 4's advisor finding)
    assert d.live_sharded_contexts() == 8
    for world in (3, 1, 5, 2, 8, 4):
        got = d.process_views_sharded(pano, yaws, pitches, 320, 180, 90.0, [0] * world)
        assert np.array_equal(got, one), world
        assert d.live_sharded_contexts() == world, (world, d.live_sharded_contexts())
    d.release_sharded()
    assert d.live_sharded_contexts() == 0
    # through the tool: a single image with --devices 0 0
    m = pkg.panorama_to_plane_pitch
    m.set_devices([0, 0])
    try:
        got = m._views_of(pano, yaws, pitches, 320, 180, 90)
    finally:
        m.set_devices(None)
    assert np.array_equal(got, one)


def test_asynchronous_copies_and_two_slot_pipeline(gpu, pkg, synth):
    """p2p_job_set_pano_async / p2p_job_get_views_async: upload of image k+1 and download of image k-1 run on their
    own streams around kernel k; the device-side ordering must give exactly the synchronous results."""
    import importlib
    d = importlib.import_module("360-to-planer-images_amd._driver")
    yaws, pitches = [0, 30, 77], [60, 120]
    panos = [synth.synth_pano(1024, 512, 3200 + i, "N") for i in range(7)]
    want = [pkg.process_views(p, yaws, pitches, 200, 120, 90) for p in panos]
    pipe = d.DevicePipeline(0)
    tickets = [pipe.submit(p, yaws, pitches, 90.0, 200, 120) for p in panos]  # 7 images through 2 slots
    for t, w in zip(tickets, want):
        assert np.array_equal(t.result(), w)
    # a geometry change re-creates the slot's job
    t = pipe.submit(panos[0], [5], [90], 90.0, 64, 48)
    assert np.array_equal(t.result(), pkg.process_views(panos[0], [5], [90], 64, 48, 90))
    pipe.close()
    # raw API: one job, asynchronous upload, run, asynchronous download, then the next image into the same job
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, 1024, 512, 1, yaws, pitches, 90, 200, 120)
    outs = []
    for p in panos[:3]:
        job.set_pano(0, p, wait=False)
        job.run()
        outs.append(job.get_views_async(0))
    job.wait()
    for o, w in zip(outs, want):
        assert np.array_equal(o, w)
    job.close()
    ctx.close()


def test_folder_walk_through_the_device_pipeline(gpu, pkg, synth, tmp_path):
    from PIL import Image
    m = pkg.panorama_to_plane_pitch
    (tmp_path / "in").mkdir()
    panos = [synth.synth_pano(512, 256, 3300 + i, "S") for i in range(5)]
    for i, p in enumerate(panos):
        Image.fromarray(np.ascontiguousarray(p[:, :, ::-1])).save(tmp_path / "in" / f"img{i}.png")
    m.main(str(tmp_path / "in"), str(tmp_path / "out"), [0, 90], [60, 90], 96, 64, num_workers=4)
    for i, p in enumerate(panos):
        want = pkg.process_views(p, [0, 90], [60, 90], 96, 64, 90)
        for yi, y in enumerate((0, 90)):
            for pi, pt in enumerate((60, 90)):
                got = np.asarray(Image.open(tmp_path / "out" / f"img{i}_96x64_yaw_{y}_pitch_{pt}.png"))[:, :, ::-1]
                assert np.array_equal(got, want[yi, pi]), (i, y, pt)


def test_one_shot_pool_is_bounded_and_survives_many_threads(gpu, pkg, synth, monkeypatch, p2p_env):
    """The reference's default fan-out is int(0.9 * cores) threads (P:304-306).  The one-shot entry points serve them
    from P2P_ONESHOT_SLOTS contexts per device: results identical, callers beyond the pool wait."""
    p2p_env("P2P_ONESHOT_SLOTS", "2")
    pano = synth.synth_pano(1024, 512, 3400, "N")
    yaws = list(range(0, 360, 10))
    want = pkg.process_views(pano, yaws, [60, 120], 96, 64)
    with ThreadPoolExecutor(max_workers=48) as ex:
        got = list(ex.map(lambda y: pkg.process_yaw_and_pitchs(pano, y, [60, 120], 96, 64), yaws))
    for yi in range(len(yaws)):
        assert np.array_equal(got[yi][0], want[yi, 0]) and np.array_equal(got[yi][1], want[yi, 1])
    gpu.release_cache()
    assert np.array_equal(pkg.process_views(pano, yaws[:2], [60], 96, 64)[1, 0], want[1, 0])


def test_plan_pass_is_timed_only_for_jobs_that_ask(gpu, synth):
    """p2p_job_plan_ms (include/p2p_hip.h): the yaw tables are always timed; the plan pass only when the job that built the plan
    had launch timing on before its first run -- two events around the pass are 10 us of a cold image's device time.  Both a
    per-view plan and a band plan (the reference CLI's default view set); the views do not depend on it."""
    pano = synth.synth_pano(2048, 1024, 7201, "S")
    for geo in ((320, 180, [0, 90, 180, 270], [60, 90, 120]), (200, 200, [0, 90, 180, 270], [30, 60, 90, 120, 150])):
        ow, oh, yaws, pitches = geo
        views = []
        for timed in (False, True):
            ctx = gpu.Context(0)   # (a context of its own: the plan comes out of no cache)
            try:
                job = gpu.Job(ctx, 2048, 1024, 1, yaws, pitches, 90, ow, oh)
                job.set_pano(0, pano)
                if timed:
                    job.time_launches(1)
                job.run()
                plan_ms, tables_ms = job.plan_ms()
                assert tables_ms > 0.0
                assert (plan_ms > 0.0) == timed, (geo, timed, plan_ms)
                views.append(job.get_views(0).copy())
                job.close()
            finally:
                ctx.close()
        assert np.array_equal(views[0], views[1])


def test_threads_build_different_plans_on_one_context(gpu, synth):
    """Several threads create jobs of DIFFERENT geometries on one context and run them at the same time: every plan pass
    counts its gather tiles in words that belong to the context (one pass at a time, csrc/p2p_host_plan.cpp) and hands the
    count to the host itself -- each job draws what the same job draws alone on a context of its own, poles (gather tiles)
    and all, on its first, second and third launch."""
    pano = synth.synth_pano(2048, 1024, 7117, "N")
    geos = [(320, 200, [60, 90, 120], 90), (256, 256, [5, 90], 100), (400, 144, [30, 150], 75), (192, 320, [90], 120),
            (333, 217, [1, 45, 179], 90), (640, 360, [60, 90, 120], 90)]
    yaws = [0, 47, 90, 201.5, 300]

    def draw(ctx, geo, launches):
        ow, oh, pitches, fov = geo
        job = gpu.Job(ctx, 2048, 1024, 1, yaws, pitches, fov, ow, oh)
        try:
            job.set_pano(0, pano)
            out = []
            for _ in range(launches):
                job.run()
                out.append(job.get_views(0).copy())
            return out
        finally:
            job.close()

    want = []
    for geo in geos:
        ctx = gpu.Context(0)
        try:
            want.append(draw(ctx, geo, 1)[0])
        finally:
            ctx.close()
    shared = gpu.Context(0)
    try:
        for rnd in range(3):
            with ThreadPoolExecutor(max_workers=len(geos)) as ex:
                got = list(ex.map(lambda g: draw(shared, g, 3), geos))
            for gi, views in enumerate(got):
                for li, v in enumerate(views):
                    assert np.array_equal(v, want[gi]), (rnd, gi, li, int((v != want[gi]).sum()))
            gpu.release_cache()   # (the next round builds every plan again)
    finally:
        shared.close()


@pytest.mark.parametrize("flags_name", ["u8", "f16"])
def test_view_mask_draws_exactly_the_wanted_views(gpu, synth, flags_name):
    """p2p_job_set_view_mask: a sparse (yaw, pitch) set -- a rank's share of one image in the view-sharded path -- in
    ONE job.  The wanted views equal the full job's bytes; the others are not touched (they keep the previous image's
    pixels); a pole pitch (gather tiles), a flickering yaw on 8192 columns (rest / table kernels) and a cleared mask
    are in."""
    pw, ph, ow, oh, fov = 8192, 4096, 320, 200, 90
    yaws, pitches = [0, 14, 33.5, 90, 200, 359], [4, 60, 90, 150]   # 14 degrees: per-column weights on 8192 columns
    flags = {"u8": 0, "f16": gpu.FLAG_PIXELS_F16}[flags_name]
    pa, pb = synth.synth_pano(pw, ph, 3700, "N"), synth.synth_pano(pw, ph, 3701, "N")
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh, flags=flags)
    full = {}
    for name, pano in (("a", pa), ("b", pb)):
        job.set_pano(0, pano)
        job.run()
        full[name] = job.get_views(0)
    rng = np.random.default_rng(7)
    mask = rng.integers(0, 2, size=(len(yaws), len(pitches))).astype(np.uint8)
    mask[1, 0] = mask[1, 2] = 1          # the flickering yaw at the pole pitch and at the horizon
    mask[0, 1] = 0
    job.set_pano(0, pa)
    job.run()                            # every view holds image a
    job.set_view_mask(mask)
    assert job.info()["n_views_wanted"] == int(mask.sum())
    job.set_pano(0, pb)
    job.run()                            # the wanted views now hold image b
    got = job.get_views(0)
    for yi in range(len(yaws)):
        for pi in range(len(pitches)):
            want = full["b" if mask[yi, pi] else "a"][yi, pi]
            assert np.array_equal(got[yi, pi], want), (flags_name, yaws[yi], pitches[pi], int(mask[yi, pi]))
            if mask[yi, pi]:
                assert np.array_equal(job.get_view(yi, pi), want)
    job.set_view_mask(None)
    job.run()
    assert np.array_equal(job.get_views(0), full["b"])
    job.close()
    ctx.close()


def test_reference_fan_out_of_230_threads_through_the_default_pool(gpu, pkg, synth):
    """P:304-306: num_workers = int(0.9 * cores) -- 230 threads on a 256-core host -- each calling
    process_yaw_and_pitchs on ONE shared panorama (P:252-265).  With the library's defaults (4 one-shot slots per
    device, one stream each) all of them are served, every view byte for byte what a single call gives, and the process
    ends up with the pool's four contexts, not with one per thread."""
    pano = synth.synth_pano(1024, 512, 3500, "N")
    yaws = list(range(0, 360, 3))          # 120 distinct yaws, each asked for by one or two of the 230 tasks
    tasks = [yaws[i % len(yaws)] for i in range(230)]
    want = pkg.process_views(pano, yaws, [60, 120], 96, 64)
    gpu.release_cache()
    with ThreadPoolExecutor(max_workers=230) as ex:
        got = list(ex.map(lambda y: pkg.process_yaw_and_pitchs(pano, y, [60, 120], 96, 64), tasks))
    for y, views in zip(tasks, got):
        yi = yaws.index(y)
        assert np.array_equal(views[0], want[yi, 0]) and np.array_equal(views[1], want[yi, 1]), y
    gpu.release_cache()


def test_release_cache_gives_device_memory_back(gpu, synth):
    """p2p_release_cache: the cached one-shot jobs, the unused tables of every live context (the caller's own
    included) and the pool's idle blocks go back to the driver -- free device memory returns to where it was."""
    def free_mb():
        return gpu.device_mem_info(0)[0] / 2**20

    pano = synth.synth_pano(4096, 2048, 3600, "S")
    gpu.release_cache()
    before = free_mb()
    ctx = gpu.Context(0)                                       # an explicit context: its caches must be reachable too
    for pitches in ([60, 90, 120], [45, 135], [30, 150]):      # three geometries -> three cached plans
        job = gpu.Job(ctx, 4096, 2048, 1, list(range(0, 360, 20)), pitches, 90, 1280, 720)
        job.set_pano(0, pano)
        job.run()
        job.get_views(0)
        job.close()
    gpu.remap_views(pano, [0, 90], [60, 120], 90, 1280, 720)   # and a one-shot slot's cached job
    held = before - free_mb()
    assert held > 150, held                                    # panoramas, views, plans and tables are being kept
    gpu.release_cache()
    after = before - free_mb()
    assert after < 32, (held, after)                           # all of it came back (the context itself holds no tables)
    ctx.close()


def test_resource_limits_of_the_header(gpu, synth, p2p_env):
    """The two resource knobs include/p2p_hip.h documents: P2P_POOL_MB = 0 -- the pool keeps no idle block, a destroyed
    job's device memory is back at the driver without p2p_release_cache -- and P2P_ONESHOT_CACHE_MAX_MB = 0 -- a one-shot
    slot keeps no job between calls (and the second call is as right as the first)."""
    def free_mb():
        return gpu.device_mem_info(0)[0] / 2**20

    pano = synth.synth_pano(4096, 2048, 3601, "S")
    gpu.release_cache()
    p2p_env("P2P_POOL_MB", "0")
    p2p_env("P2P_PLAN_CACHE", "0")                             # (the context would keep the plan for the next job)
    before = free_mb()
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, 4096, 2048, 1, list(range(0, 360, 30)), [60, 90, 120], 90, 1280, 720)
    job.set_pano(0, pano)
    job.run()
    first = job.get_views(0)
    assert before - free_mb() > 100                            # 25 MB of panorama, 100 MB of views, the plan
    job.close()
    assert before - free_mb() < 32, before - free_mb()         # nothing idles in the pool
    ctx.close()
    p2p_env("P2P_ONESHOT_CACHE_MAX_MB", "0")
    a = gpu.remap_views(pano, list(range(0, 360, 30)), [60, 90, 120], 90, 1280, 720)
    held = before - free_mb()
    b = gpu.remap_views(pano, list(range(0, 360, 30)), [60, 90, 120], 90, 1280, 720)
    assert held < 32 and np.array_equal(a, b) and np.array_equal(a, first), held
    gpu.release_cache()


def test_process_exit_with_live_caches_from_pool_threads(gpu, tmp_path):
    """Cached one-shot jobs, streams and page-locked blocks are alive when the interpreter exits -- from pool
    threads that finished, from a daemon thread that never will, and on sys.exit from the main thread.  Nothing is
    torn down through HIP after the runtime's own shutdown: the process exits 0 without a crash."""
    import subprocess
    import sys
    root = str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    code = r'''
import importlib, sys, threading, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, %r)
import numpy as np
p = importlib.import_module("360-to-planer-images_amd")
pano = np.random.default_rng(0).integers(0, 256, (256, 512, 3), dtype=np.uint8)
ex = ThreadPoolExecutor(max_workers=6)
list(ex.map(lambda y: p.process_yaw_and_pitchs(pano, y, [90], 64, 48), range(0, 360, 30)))
def forever():
    while True:
        p.process_yaw_and_pitchs(pano, 7, [60], 64, 48)
threading.Thread(target=forever, daemon=True).start()
time.sleep(0.2)
print("alive", flush=True)
sys.exit(int(sys.argv[1]))
''' % root
    for rc in (0, 3):
        r = subprocess.run([sys.executable, "-c", code, str(rc)], capture_output=True, text=True, timeout=300)
        assert r.returncode == rc, (r.returncode, r.stderr[-1500:])
        assert "alive" in r.stdout
        assert "Segmentation" not in r.stderr and "core dumped" not in r.stderr


def test_no_device_or_host_memory_drift_over_many_images(gpu):
    """tools/soak_leak.py: one-shot calls from several threads, the two-slot pipeline, the view-sharded driver and
    the float path over changing geometries, round after round -- free device memory and the resident set stay put
    once the caches have filled."""
    import subprocess
    import sys
    root = __import__("pathlib").Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "soak_leak.py"), "--rounds", "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "drift after round" in r.stdout
