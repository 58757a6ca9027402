"""GPU, BASELINE.json's configurations at their full sizes.
  config 2  every one of the 36 views against the oracle (caller-map mode bit-exact, fused within +-1), the
            roll property of whole-column yaws, fused == caller-map mode fed the kernel's own coordinates;
  config 3  one GPU's share of the 64-panorama batch (8 panoramas resident, 288 views in one launch): two
            panoramas x two yaws against the oracle, the rest self-consistent;
  config 4  16384x8192 -> 4096x4096, FOV 60, NOISE panorama: pitch 30 (pole in view: split tiles, direct
            gathers) and pitch 90 for one yaw against the oracle, plus the constant-colour property;
  config 5  the 360-yaw sweep at 8K: four yaws (whole-column, fractional, flickering fraction) against the
            oracle, and the float16 pixel path against the float32 one over the whole sweep.
The oracle runs one task per yaw on the host's cores (the reference's own parallelism, P:252-265)."""
import numpy as np
import pytest

from _util import coords_to_maps, diff_stats, oracle_maps, oracle_views, oracle_views_threaded
from oracle import maps

pytestmark = pytest.mark.gpu

CFG2 = dict(pw=8192, ph=4096, ow=1920, oh=1080, fov=90, pitches=[60, 90, 120])


@pytest.fixture(scope="module")
def pano8k(synth):
    return synth.synth_pano(CFG2["pw"], CFG2["ph"], 1000, "S")


def test_cfg2_all_36_views_vs_oracle(gpu, pkg, pano8k):
    c = CFG2
    yaws = list(range(0, 360, 30))
    want = oracle_views_threaded(pano8k, yaws, c["pitches"], c["ow"], c["oh"], c["fov"])
    rows, U, V = oracle_maps(yaws, c["pitches"], c["ow"], c["oh"], c["pw"], c["ph"], c["fov"])
    exact = gpu.remap_views_maps(pano8k, rows, U, V)
    bad = np.argwhere(exact != want)
    assert bad.size == 0, (len(bad), bad[:4].tolist())  # 74.6 Mpix x 3 channels, every byte
    fused = pkg.process_views(pano8k, yaws, c["pitches"], c["ow"], c["oh"], c["fov"])
    mx, gt1, anyd = diff_stats(fused, want)
    print("cfg2 all views, fused vs oracle: max %d, >1: %.3g, any: %.3g" % (mx, gt1, anyd))
    assert mx <= 1


def test_cfg2_whole_column_shift_equals_roll(gpu, pkg, pano8k):
    c = CFG2
    a = pkg.process_views(pano8k, [45, 180], c["pitches"], c["ow"], c["oh"], c["fov"])
    for k, yaw in enumerate((45, 180)):
        shift = yaw * c["pw"] // 360
        b = pkg.process_views(np.roll(pano8k, -shift, axis=1), [0], c["pitches"], c["ow"], c["oh"], c["fov"])
        assert np.array_equal(a[k], b[0])


def test_cfg2_all_36_views_self_consistent(gpu, pkg, pano8k):
    c = CFG2
    yaws = list(range(0, 360, 30))
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, c["pw"], c["ph"], 1, yaws, c["pitches"], c["fov"], c["ow"], c["oh"])
    job.set_pano(0, pano8k)
    job.run()
    fused, coords = job.get_views(0), job.get_coords()
    job.close()
    U, V = zip(*(coords_to_maps(coords[p]) for p in range(3)))
    rows = np.stack([maps.yaw_column_table(c["pw"], y) for y in yaws])
    again = gpu.remap_views_maps(pano8k, rows, np.stack(U), np.stack(V))
    assert np.array_equal(fused, again)  # 74.6 Mpix, every byte
    ctx.close()


def test_constant_panorama_constant_views_cfg4_size(gpu, pkg):
    # config 4 geometry (16384x8192 -> 4096x4096, FOV 60) on a few views incl. the pole-containing pitch 30
    pano = np.empty((8192, 16384, 3), np.uint8)
    pano[:] = (7, 130, 251)
    v = pkg.process_views(pano, [5, 200], [30, 90], 4096, 4096, 60)
    assert (v == np.array([7, 130, 251], np.uint8)).all()


def test_cfg3_one_gpu_share_of_the_64_panorama_batch(gpu, synth):
    """8 panoramas (seeds as bench.py deals them to rank 0 of 8) x 36 views resident, one launch."""
    c = CFG2
    yaws = list(range(0, 360, 30))
    n = 8
    panos = [synth.synth_pano(c["pw"], c["ph"], 1000 + i, "N") for i in range(n)]
    rows, U, V = oracle_maps(yaws, c["pitches"], c["ow"], c["oh"], c["pw"], c["ph"], c["fov"])
    ctx = gpu.Context(0)
    fused = gpu.Job(ctx, c["pw"], c["ph"], n, yaws, c["pitches"], c["fov"], c["ow"], c["oh"])
    exact = gpu.Job(ctx, c["pw"], c["ph"], n, yaws, c["pitches"], c["fov"], c["ow"], c["oh"])
    assert fused.info()["tile_w"] == 128   # 1.8 GB of views from several resident panoramas: the library's rule (choose_shape)
    for i, p in enumerate(panos):
        fused.set_pano(i, p)
        exact.set_pano(i, p)
    fused.run()
    coords = fused.get_coords()
    Uk, Vk = zip(*(coords_to_maps(coords[p]) for p in range(len(c["pitches"]))))
    # caller-map mode with the ORACLE's maps: bit-exact against the oracle on a sample
    exact.set_maps(rows, U, V)
    exact.run()
    for i, ysel in ((0, [0, 5]), (7, [3, 11])):
        got = exact.get_views(i)
        want = oracle_views_threaded(panos[i], [yaws[y] for y in ysel], c["pitches"], c["ow"], c["oh"], c["fov"])
        for k, y in enumerate(ysel):
            assert np.array_equal(got[y], want[k]), (i, yaws[y])
    # caller-map mode with the fused kernel's own coordinates: every one of the 288 views, byte for byte
    exact.set_maps(rows, np.stack(Uk), np.stack(Vk))
    exact.run()
    for i in range(n):
        assert np.array_equal(fused.get_views(i), exact.get_views(i)), i
    fused.close()
    exact.close()
    ctx.close()


def test_cfg4_noise_panorama_pole_and_horizon_vs_oracle(gpu, synth):
    pw, ph, ow, oh, fov = 16384, 8192, 4096, 4096, 60
    pano = synth.synth_pano(pw, ph, 1000, "N")
    yaws, pitches = [35], [30, 90]
    want = oracle_views_threaded(pano, yaws, pitches, ow, oh, fov)
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    got = gpu.remap_views_maps(pano, rows, U, V)
    for pi, pitch in enumerate(pitches):
        bad = np.argwhere(got[0, pi] != want[0, pi])
        assert bad.size == 0, (pitch, len(bad), bad[:4].tolist())


def test_cfg4_five_pitch_job_sampled_views_vs_oracle(gpu, synth):
    """Config 4 as ONE job with all five pitch views (the shape that behaves differently: five plans resident, pole tiles
    of pitch 30 / 150 drawn by the gather kernel beside the LDS-scheme tiles, chunks of pairs per pitch view): a handful
    of (yaw, pitch) views of the job, byte for byte against the oracle on a noise panorama.  10 yaws keep the job's
    5 GB of views and the CPU oracle's time in bounds; the 72-yaw job itself runs in bench.py."""
    pw, ph, ow, oh, fov = 16384, 8192, 4096, 4096, 60
    pano = synth.synth_pano(pw, ph, 1004, "N")
    yaws, pitches = list(range(0, 360, 36)), [30, 60, 90, 120, 150]
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
    assert job.info()["tile_w"] == 64   # 2.5 GB of views: below the 4 GB from which a job's views stream in 128-wide tiles
    job.set_pano(0, pano)
    job.set_maps(rows, U, V)
    job.run()
    got = job.get_views(0)
    job.close()
    ctx.close()
    yi = 3  # yaw 108: a fractional shift on 16384 columns
    want = oracle_views_threaded(pano, [yaws[yi]], pitches, ow, oh, fov)
    for pi, pitch in enumerate(pitches):
        bad = np.argwhere(got[yi, pi] != want[0, pi])
        assert bad.size == 0, (pitch, len(bad), bad[:4].tolist())
    # and a whole-column shift of the same job (stage 1 is a copy: the other loop of the kernels) on the horizon view
    assert np.array_equal(got[0, 2], oracle_views_threaded(pano, [0], [90], ow, oh, fov)[0, 0])


def test_cfg4_in_the_shape_it_ships_in_sampled_views_vs_oracle(gpu, synth):
    """Config 4 exactly as bench.py launches it: 72 yaws x 5 pitches of a 16K noise panorama, 18 GB of views in ONE job
    -- which makes choose_shape pick the 128-wide tiles, cuts the 72 yaws into three chunks of 24 pairs that ONE workgroup
    per tile draws in turn (565 MB of plan tables: read once per tile), puts the main kernel's tiles in list order and
    adds the table-prefetch workgroups, one per 24 entries.
    The test names that shape (p2p_job_get_info) and checks six views byte for byte against the oracle: one or two per
    chunk of pairs, both polar pitches (gather tiles), the horizon, whole-column and fractional yaws.  Views come back
    one at a time (p2p_job_get_view): nobody holds 18 GB on the host."""
    pw, ph, ow, oh, fov = 16384, 8192, 4096, 4096, 60
    pano = synth.synth_pano(pw, ph, 1004, "N")
    yaws, pitches = list(range(0, 360, 5)), [30, 60, 90, 120, 150]
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
    info = job.info()
    assert (info["tile_w"], info["tile_h"]) == (128, 16), info
    assert info["pair_chunks"] >= 2 and info["pairs_per_block"] * info["pair_chunks"] >= 72, info   # several chunks of pairs per tile
    assert info["list_order"] == 1 and info["prefetch_lead"] > 0, info
    assert info["chunks_per_workgroup"] == info["pair_chunks"] and info["main_group"] == 24, info  # one workgroup per tile draws all three chunks
    job.set_pano(0, pano)
    job.set_maps(rows, U, V)
    job.run()
    assert job.info()["n_gather_tiles"] > 0   # the pole tiles of pitch 30 / 150
    # (yaw index, pitch index): chunk 0 = yaws 0..23, chunk 1 = 24..47, chunk 2 = 48..71; 16384 columns: 45-degree
    # multiples are whole-column shifts (stage 1 copies), everything else blends
    sample = [(0, 2), (7, 0), (27, 1), (31, 4), (50, 0), (71, 3)]
    got = {v: job.get_view(*v) for v in sample}
    job.close()
    ctx.close()
    for yi in sorted({y for y, _ in sample}):
        pis = [p for y, p in sample if y == yi]
        want = oracle_views_threaded(pano, [yaws[yi]], [pitches[p] for p in pis], ow, oh, fov)
        for k, pi in enumerate(pis):
            bad = np.argwhere(got[(yi, pi)] != want[0, k])
            assert bad.size == 0, (yaws[yi], pitches[pi], len(bad), bad[:4].tolist())


@pytest.mark.parametrize("tile_shape", ["64", "128"])
def test_cfg3_share_in_both_tile_shapes_vs_oracle(gpu, synth, p2p_env, tile_shape):
    """One GPU's share of config 3 (8 resident panoramas x 36 views, 1.8 GB: the library's own rule picks 128-wide tiles,
    grid order, chunks outermost) with each tile shape forced in turn; two panoramas x two yaws x three pitches byte
    for byte against the oracle in both."""
    c = CFG2
    yaws = list(range(0, 360, 30))
    n = 8
    p2p_env("P2P_TILE_SHAPE", tile_shape)
    panos = [synth.synth_pano(c["pw"], c["ph"], 1000 + i, "N") for i in range(n)]
    rows, U, V = oracle_maps(yaws, c["pitches"], c["ow"], c["oh"], c["pw"], c["ph"], c["fov"])
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, c["pw"], c["ph"], n, yaws, c["pitches"], c["fov"], c["ow"], c["oh"])
    assert job.info()["tile_w"] == int(tile_shape) and job.info()["list_order"] == 0
    for i, p in enumerate(panos):
        job.set_pano(i, p)
    job.set_maps(rows, U, V)
    job.run()
    for i, ysel in ((1, [2, 9]), (6, [0, 7])):
        want = oracle_views_threaded(panos[i], [yaws[y] for y in ysel], c["pitches"], c["ow"], c["oh"], c["fov"])
        for k, y in enumerate(ysel):
            for pi in range(len(c["pitches"])):
                assert np.array_equal(job.get_view(y, pi, index=i), want[k, pi]), (tile_shape, i, yaws[y], pi)
    job.close()
    ctx.close()


def test_cfg5_yaw_sweep_at_8k(gpu, pkg, pano8k):
    c = CFG2
    pitches = [90]
    sample = [0, 1, 14, 200]  # whole-column, fractional, flickering fraction (per-column weights), fractional
    want = oracle_views_threaded(pano8k, sample, pitches, c["ow"], c["oh"], c["fov"])
    ctx = gpu.Context(0)
    sweep = list(range(360))
    job = gpu.Job(ctx, c["pw"], c["ph"], 1, sweep, pitches, c["fov"], c["ow"], c["oh"])
    job.set_pano(0, pano8k)
    job.run()
    got = job.get_views(0)
    for k, y in enumerate(sample):
        mx, gt1, _ = diff_stats(got[y], want[k])
        assert mx <= 1, (y, mx, gt1)
    rows, U, V = oracle_maps(sweep, pitches, c["ow"], c["oh"], c["pw"], c["ph"], c["fov"])
    job.set_maps(rows, U, V)
    job.run()
    got = job.get_views(0)
    for k, y in enumerate(sample):
        assert np.array_equal(got[y], want[k]), y
    del got
    job.close()
    # float16 pixel path against the float32 one over the whole sweep (config 5's tolerance: 1 level)
    worst = 0
    for lo in range(0, 360, 90):
        ys = sweep[lo:lo + 90]
        res = []
        for flag in (gpu.FLAG_PIXELS_F32, gpu.FLAG_PIXELS_F16):
            j = gpu.Job(ctx, c["pw"], c["ph"], 1, ys, pitches, c["fov"], c["ow"], c["oh"], flags=flag)
            j.set_pano(0, pano8k)
            j.run()
            res.append(j.get_views(0))
            j.close()
        worst = max(worst, int(np.abs(res[0].astype(np.int16) - res[1].astype(np.int16)).max()))
    assert worst <= 1, worst
    ctx.close()


def test_reference_cli_default_view_set_on_an_8k_noise_panorama(gpu, pkg, synth):
    """The reference CLI's defaults (P:412-437): 800 x 800, FOV 90, yaw 0 / 90 / 180 / 270, pitch 30 / 60 / 90 / 120
    / 150 -- 2.56 source pixels per output pixel (every tile splits into small pieces) and a pole inside the
    pitch 30 / 150 views (the direct-gather kernel draws its surroundings).  All 20 views byte for byte."""
    pw, ph, ow, oh, fov = 8192, 4096, 800, 800, 90
    yaws, pitches = [0, 90, 180, 270], [30, 60, 90, 120, 150]
    pano = synth.synth_pano(pw, ph, 4242, "N")
    want = oracle_views_threaded(pano, yaws, pitches, ow, oh, fov)
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    exact = gpu.remap_views_maps(pano, rows, U, V)
    bad = np.argwhere(exact != want)
    assert bad.size == 0, (len(bad), bad[:4].tolist())
    smooth = synth.synth_pano(pw, ph, 4243, "S")
    fused = pkg.process_views(smooth, yaws, pitches, ow, oh, fov)
    mx, gt1, anyd = diff_stats(fused, oracle_views_threaded(smooth, yaws, pitches, ow, oh, fov))
    print("CLI default view set, fused vs oracle: max %d, >1: %.3g, any: %.3g" % (mx, gt1, anyd))
    assert mx <= 1
