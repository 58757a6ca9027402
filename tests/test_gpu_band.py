"""GPU parity of the source-band tiles (csrc/p2p_plan.hip: the band passes; csrc/p2p_views.hip: remap_views_band_kernel):
tiles that are rectangles of the SOURCE, with the 4-pixel groups of every pitch view binned to them on the device.  The
bytes must be those of the CPU restatement of the reference's two cv2.remap stages (P:181-221), whatever the cells, the
tile shape, the chunking of the pairs or a view mask do to which workgroup draws which group."""
import itertools

import numpy as np
import pytest

from _util import oracle_maps, oracle_views, oracle_views_threaded, diff_stats

pytestmark = pytest.mark.gpu


def _run(gpu, panos, yaws, pitches, ow, oh, fov, maps, mask=None):
    ph, pw = panos[0].shape[:2]
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, pw, ph, len(panos), yaws, pitches, fov, ow, oh)
        if maps is not None:
            job.set_maps(*maps)
        for i, p in enumerate(panos):
            job.set_pano(i, p)
        if mask is not None:
            job.set_view_mask(mask)
        job.run()
        info = job.info()
        out = [job.get_views(i) for i in range(len(panos))]
        job.close()
        return out, info
    finally:
        ctx.close()


@pytest.mark.parametrize("tile_shape", ["0", "64", "128"])  # (0: the library's choice -- the band shape, p2p_views_band.hip)
@pytest.mark.parametrize("n_panos", [1, 2])
def test_band_tiles_draw_the_oracles_bytes(gpu, synth, p2p_env, n_panos, tile_shape):
    """Plain-shift yaws (whole-column and fractional), a pole in view (those tiles stay with the gather kernel), a view
    width that is not divisible by 4, several chunks of pairs, the split tail, every cell geometry."""
    pw, ph, ow, oh, fov = 2048, 1024, 333, 210, 90
    yaws = [0, 33, 90, 123.4, 180, 200, 270, 301, 359]
    pitches = [8, 60, 90, 150]
    panos = [synth.synth_pano(pw, ph, 5200 + i, "N") for i in range(n_panos)]
    maps = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    want = [oracle_views(p, yaws, pitches, ow, oh, fov) for p in panos]
    p2p_env("P2P_PLAN_CACHE", "0")
    p2p_env("P2P_TILE_SHAPE", tile_shape)
    p2p_env("P2P_BAND", "1")
    for (bh, cw), ppb, merge in itertools.product(((16, 8), (4, 8), (8, 32), (24, 4)), ("0", "3"), ("1", "0")):
        p2p_env("P2P_BAND_BH", str(bh))
        p2p_env("P2P_BAND_CW", str(cw))
        p2p_env("P2P_PAIRS_PER_BLOCK", ppb)
        p2p_env("P2P_MERGE_GATHER", merge)  # the gather tiles as the first workgroups of the band kernel's launch, or as their own
        got, info = _run(gpu, panos, yaws, pitches, ow, oh, fov, maps)
        assert info["band_tiles"] > 0 and 0 < info["n_gather_tiles"] < info["n_tiles"], info
        for i in range(n_panos):
            bad = np.argwhere(got[i] != want[i])
            assert bad.size == 0, (tile_shape, bh, cw, ppb, merge, i, len(bad), bad[:3])


def test_band_tiles_with_a_view_mask(gpu, synth, p2p_env):
    """p2p_job_set_view_mask on a band plan: the lanes of a tile belong to several pitch views -- a wanted view gets the
    full job's bytes, an unwanted one is not touched."""
    pw, ph, ow, oh, fov = 2048, 1024, 480, 270, 90
    yaws, pitches = [0, 30, 77, 180, 270], [45, 90, 120]
    pano = synth.synth_pano(pw, ph, 5300, "N")
    maps = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    want = oracle_views(pano, yaws, pitches, ow, oh, fov)
    p2p_env("P2P_PLAN_CACHE", "0")
    p2p_env("P2P_BAND", "1")
    rng = np.random.default_rng(5)
    for _ in range(3):
        mask = (rng.random((len(yaws), len(pitches))) < 0.5).astype(np.uint8)
        mask[0, 0] = 1
        (got,), info = _run(gpu, [pano], yaws, pitches, ow, oh, fov, maps, mask)
        assert info["band_tiles"] > 0 and info["n_views_wanted"] == int(mask.sum())
        m = mask.astype(bool)
        assert np.array_equal(got[m], want[m])


def test_a_flickering_yaw_turns_the_band_plan_off(gpu, synth, p2p_env):
    """Band tiles draw plain-shift yaws only: a job with a yaw whose table has per-column weights keeps the per-view
    tiles (and its rest kernel), also when P2P_BAND=1 asks for band tiles; p2p_job_set_yaws back to plain yaws
    rebuilds the plan as a band plan."""
    pw, ph, ow, oh, fov = 8192, 4096, 320, 180, 90
    pano = synth.synth_pano(pw, ph, 5400, "N")
    p2p_env("P2P_BAND", "1")
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, pw, ph, 1, [0, 14], [90], fov, ow, oh)  # yaw 14 flickers on 8192 columns
        job.set_pano(0, pano)
        job.run()
        assert job.info()["n_odd_yaws"] == 1 and job.info()["band_tiles"] == 0
        a = job.get_views(0)
        want = oracle_views(pano, [0, 14], [90], ow, oh, fov)
        mx, gt1, _ = diff_stats(a, want)
        assert mx <= 16 and gt1 < 5e-4
        job.set_yaws([0, 30])
        job.run()
        assert job.info()["n_odd_yaws"] == 0 and job.info()["band_tiles"] > 0
        job.close()
    finally:
        ctx.close()


def test_reference_cli_default_view_set_through_band_tiles_at_8k(gpu, pkg, synth, p2p_env):
    """The reference CLI's defaults (P:412-437) at 8K are what the library's own rule sends through band tiles
    (2.56 source pixels per output pixel, five pitch views over every source rectangle).  Caller maps (the oracle's own)
    with band tiles forced: all 20 views byte for byte; device maps under the library's rule: band tiles chosen, +-1."""
    pw, ph, ow, oh, fov = 8192, 4096, 800, 800, 90
    yaws, pitches = [0, 90, 180, 270], [30, 60, 90, 120, 150]
    pano = synth.synth_pano(pw, ph, 4242, "N")
    want = oracle_views_threaded(pano, yaws, pitches, ow, oh, fov)
    maps = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    p2p_env("P2P_BAND", "1")
    (got,), info = _run(gpu, [pano], yaws, pitches, ow, oh, fov, maps)
    assert info["band_tiles"] > 3000 and info["n_gather_tiles"] < 400, info
    bad = np.argwhere(got != want)
    assert bad.size == 0, (len(bad), bad[:4].tolist())
    p2p_env("P2P_BAND", "-1")
    smooth = synth.synth_pano(pw, ph, 4243, "S")
    (fused,), info = _run(gpu, [smooth], yaws, pitches, ow, oh, fov, None)
    assert info["band_tiles"] > 3000, info  # the library's rule
    mx, gt1, anyd = diff_stats(fused, oracle_views_threaded(smooth, yaws, pitches, ow, oh, fov))
    assert mx <= 1, (mx, gt1, anyd)


def test_band_kernel_draws_the_same_bytes_every_time(gpu, synth, p2p_env):
    """The band kernel's workgroups alternate between two LDS buffers with ONE barrier per pair: a wave that wrote past
    the end of its buffer would change pixels that slower waves are still reading from the other -- in a few runs of
    many (buffers of 1000 items, not whole waves of them, did exactly that: 3 runs of 14).  The reference CLI's default
    set at 8K, cold (new context, new plan) and warm, every run against the per-view tiles' bytes
    (tests/fuzz/band_race.py is the long form)."""
    pw, ph, ow, oh, fov = 8192, 4096, 800, 800, 90
    yaws, pitches = [0, 90, 180, 270], [30, 60, 90, 120, 150]
    pano = synth.synth_pano(pw, ph, 4242, "N")
    p2p_env("P2P_BAND", "0")
    (want,), info = _run(gpu, [pano], yaws, pitches, ow, oh, fov, None)
    assert info["band_tiles"] == 0
    p2p_env("P2P_BAND", "-1")
    for _ in range(12):
        (got,), info = _run(gpu, [pano], yaws, pitches, ow, oh, fov, None)
        assert info["band_tiles"] > 3000 and info["lds_items_cap"] == 960, info
        assert np.array_equal(got, want)
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
        job.set_pano(0, pano)
        for _ in range(60):
            job.run()
            assert np.array_equal(job.get_views(0), want)
        job.close()
    finally:
        ctx.close()


def test_config_2_keeps_the_per_view_tiles(gpu, synth):
    """BASELINE config 2 reads 1.07 source pixels per output pixel: the library's rule leaves it with the per-view
    tiles (band tiles there: byte-equal, fewer instructions, and 131 us against 83 -- their ragged row ends need
    write-back stores)."""
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, 8192, 4096, 1, list(range(0, 360, 30)), [60, 90, 120], 90, 1920, 1080)
        assert job.info()["band_tiles"] == 0
        job.close()
    finally:
        ctx.close()
