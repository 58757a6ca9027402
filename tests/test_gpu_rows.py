"""p2p_job_set_rows / p2p_job_get_view_rows: one image's ROWS shared out to several GPUs (fewer images than
GPUs).  Every rank's job draws all the views, a band of whole tile rows of each; the plan is made for the band (tiles
outside it are nobody's).  Whatever the band, its rows are the rows of the whole job -- and, with the reference's own
maps, the oracle's bytes (P:181-221)."""
import numpy as np
import pytest

from _util import oracle_maps, oracle_views

pytestmark = pytest.mark.gpu


def _whole_and_bands(gpu, pano, yaws, pitches, ow, oh, fov, bands, maps=None, mask=None):
    ph, pw = pano.shape[:2]
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
        if maps is not None:
            job.set_maps(*maps)
        if mask is not None:
            job.set_view_mask(mask)
        job.set_pano(0, pano)
        job.run()
        whole = job.get_views(0).copy()
        info_whole = job.info()
        got = np.zeros_like(whole)
        infos = []
        for r0, r1 in bands:
            job.set_rows(r0, r1)
            job.run()
            infos.append(job.info())
            for y in range(len(yaws)):
                for p in range(len(pitches)):
                    if y % 2:
                        got[y, p, r0:r1] = job.get_view_rows(y, p, r0, r1)
                    elif ow % 4 == 0:
                        job.get_view_rows_async(y, p, r0, r1, got[y, p, r0:r1])
                    else:
                        got[y, p, r0:r1] = job.get_view_rows(y, p, r0, r1)
            job.wait()
        job.set_rows(0, oh)
        job.run()
        again = job.get_views(0).copy()
        job.close()
        return whole, got, again, info_whole, infos
    finally:
        ctx.close()


@pytest.mark.parametrize("tile_shape", ["0", "128"])
def test_bands_of_rows_are_the_whole_jobs_rows(gpu, synth, p2p_env, tile_shape):
    """Per-view tiles (LDS scheme and, around the poles, the gather kernel), fractional and whole-column yaws, a view
    height that is not a multiple of 16, caller maps: the oracle's bytes, band by band."""
    p2p_env("P2P_TILE_SHAPE", tile_shape)
    pw, ph, ow, oh, fov = 2048, 1024, 320, 200, 90
    yaws, pitches = [0, 33, 90, 200.5], [20, 90, 140]
    pano = synth.synth_pano(pw, ph, 6100, "N")
    maps = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    want = oracle_views(pano, yaws, pitches, ow, oh, fov)
    bands = [(0, 48), (48, 64), (64, 160), (160, 200)]
    whole, got, again, info_whole, infos = _whole_and_bands(gpu, pano, yaws, pitches, ow, oh, fov, bands, maps)
    assert np.array_equal(whole, want)
    assert np.array_equal(got, want)
    assert np.array_equal(again, want)
    # a band's plan holds the band's tiles only
    assert all(i["n_gather_tiles"] <= info_whole["n_gather_tiles"] for i in infos)
    assert sum(i["n_gather_tiles"] for i in infos) == info_whole["n_gather_tiles"]


def test_bands_of_rows_with_device_maps_band_tiles_and_a_view_mask(gpu, synth, p2p_env):
    """Device maps (what the sharded driver runs), an odd width (the synchronous download), source-band tiles forced, a
    view mask on top: every band equals the whole job's rows."""
    pw, ph, ow, oh, fov = 2048, 1024, 318, 180, 90
    yaws, pitches = [0, 30, 77, 180], [60, 90, 120]
    pano = synth.synth_pano(pw, ph, 6101, "N")
    bands = [(0, 64), (64, 128), (128, 180)]
    for band_opt in ("0", "1"):
        p2p_env("P2P_BAND", band_opt)
        whole, got, again, _, infos = _whole_and_bands(gpu, pano, yaws, pitches, ow, oh, fov, bands)
        assert np.array_equal(got, whole), band_opt
        assert np.array_equal(again, whole), band_opt
        if band_opt == "1":
            assert all(i["band_tiles"] > 0 for i in infos), infos
    mask = np.zeros((4, 3), np.uint8)
    mask[0, 0] = mask[1, 0] = mask[2, 1] = mask[3, 2] = 1
    p2p_env("P2P_BAND", "-1")
    whole, got, again, _, _ = _whole_and_bands(gpu, pano, yaws, pitches, ow, oh, fov, bands, mask=mask)
    m = mask.astype(bool)
    assert np.array_equal(got[m], whole[m]) and np.array_equal(again[m], whole[m])


def test_row_ranges_are_whole_tile_rows(gpu, synth):
    pano = synth.synth_pano(512, 256, 6102, "N")
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, 512, 256, 1, [0], [90], 90, 128, 100)
        job.set_pano(0, pano)
        for bad in ((8, 32), (0, 40), (32, 32), (48, 16), (-16, 16), (0, 112)):
            with pytest.raises(gpu.P2PError):
                job.set_rows(*bad)
        job.run()
        assert job.get_coords().shape == (1, 100, 128, 2)
        job.set_rows(96, 100)  # (the last tile row is short)
        with pytest.raises(gpu.P2PError):
            job.get_coords()  # (the plan those coordinates belonged to is gone: run first)
        job.run()
        assert job.get_coords().shape == (1, 100, 128, 2)
        with pytest.raises(gpu.P2PError):
            job.get_view_rows(0, 0, 90, 101)
        assert job.get_view_rows(0, 0, 96, 100).shape == (4, 128, 3)
        job.close()
    finally:
        ctx.close()


def test_rows_through_the_sharded_driver_incl_the_float_pixel_paths(gpu, synth):
    """_driver.process_views_sharded, one image on three contexts of one device: a band of rows of every view per context
    ("rows") and whole views per context ("views") stitch to the single-context result -- uint8 and both float pixel paths."""
    import importlib
    d = importlib.import_module("360-to-planer-images_amd._driver")
    pano = synth.synth_pano(2048, 1024, 3100, "S")
    yaws, pitches = [0, 33.3, 90, 200], [40, 90, 150]
    try:
        for flags in (0, gpu.FLAG_PIXELS_F32, gpu.FLAG_PIXELS_F16):
            one = d.process_views_sharded(pano, yaws, pitches, 320, 200, 90.0, [0], flags=flags)
            for how in ("rows", "views"):
                got = d.process_views_sharded(pano, yaws, pitches, 320, 200, 90.0, [0, 0, 0], flags=flags, how=how)
                assert np.array_equal(got, one), (flags, how)
    finally:
        d.release_sharded()
