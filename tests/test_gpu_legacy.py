"""GPU: the legacy tool (app/legacy/panorama_to_plane.py, "L") end to end -- combined-rotation maps from
rot_map_kernel against the reference-pinned restatement, the three interpolate_color methods bit-exact against
the CPU restatement of cv2.remap, and the legacy CLI's files."""
import importlib

import numpy as np
import pytest

from oracle import cpu_ref, maps

pytestmark = pytest.mark.gpu

MODES = [cpu_ref.BORDER_CONSTANT, cpu_ref.BORDER_REPLICATE, cpu_ref.BORDER_REFLECT, cpu_ref.BORDER_WRAP,
         cpu_ref.BORDER_REFLECT_101]


@pytest.fixture(scope="module")
def legacy(pkg):
    return importlib.import_module("360-to-planer-images_amd.panorama_to_plane")


@pytest.mark.parametrize("args", [
    (64, 48, 90, 77, 60, 256, 128), (64, 48, 60, 300, 30, 256, 128), (64, 48, 90, 45, 179, 256, 128),
    (1000, 1500, 90, 60, 90, 8192, 4096), (1000, 1500, 90, 240, 90, 8192, 4096), (800, 600, 100, 200, 120, 4096, 2048),
])
def test_combined_rotation_map_1e5(gpu, legacy, args):
    W, H, fov, yaw, pitch, pw, ph = args
    legacy.precompute_mapping.cache_clear()
    fr, yr, pr = float(np.radians(fov)), float(np.radians(yaw)), float(np.radians(pitch))
    U, V = legacy.precompute_mapping(W, H, fr, yr, pr, pw, ph)
    assert legacy.precompute_mapping(W, H, fr, yr, pr, pw, ph)[0] is U          # lru_cache, as L:47
    Ur, Vr = maps.legacy_map(W, H, fr, yr, pr, pw, ph)
    assert U.dtype == np.float32 and U.shape == (H, W)
    ok = ~(np.isnan(Vr) | np.isnan(V))
    assert ok.mean() > 0.9999
    th, thr = V[ok].astype(np.float64) * np.pi / ph, Vr[ok].astype(np.float64) * np.pi / ph
    assert np.abs(np.cos(th) - np.cos(thr)).max() <= 1e-5
    sin_t = np.maximum(np.sin(thr), 1e-6)          # conditioning of arccos / arctan2, as for the pitch map
    assert (np.abs(V[ok] - Vr[ok]) <= 1e-5 * np.abs(Vr[ok]) + 1e-5 / sin_t * ph / np.pi).all()
    dU = np.abs(U[ok].astype(np.float64) - Ur[ok])
    dU = np.minimum(dU, pw - 1 - dU)
    assert (dU <= 1e-5 * np.abs(Ur[ok]) + 1e-5 / sin_t * pw / (2 * np.pi)).all()
    print("bit-equal: U %.4f V %.4f" % ((U == Ur)[ok].mean(), (V == Vr)[ok].mean()))


@pytest.mark.parametrize("method,code", [("nearest", 0), ("bilinear", 1), ("bicubic", 2), ("lanczos?", 1)])
def test_interpolate_color_methods(gpu, legacy, synth, method, code):
    pano = synth.synth_pano(1024, 512, 2100, "N")
    U, V = maps.legacy_map(320, 200, float(np.radians(90)), float(np.radians(77)), float(np.radians(60)), 1024, 512)
    got = legacy.interpolate_color(U, V, pano, method)
    assert np.array_equal(got, cpu_ref.remap(pano, U, V, cpu_ref.BORDER_REFLECT, None, interpolation=code))
    if method == "bilinear":
        assert np.array_equal(got, legacy.panorama_to_plane(pano, U, V))


@pytest.mark.parametrize("interp", [0, 2])
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_nearest_and_cubic_all_borders(gpu, interp, mode, cn):
    rng = np.random.default_rng(400 + 100 * interp + 10 * mode + cn)
    img = rng.integers(0, 256, size=(37, 53, cn), dtype=np.uint8)
    U = rng.uniform(-70, 130, size=(64, 80)).astype(np.float32)
    V = rng.uniform(-50, 90, size=(64, 80)).astype(np.float32)
    U[0, :8] = [np.nan, 0.0, 52.0, 52.5, -1.0, 1e9, 0.5, 1.5]
    V[0, :8] = [3.0, np.nan, 36.0, 36.5, -0.5, -1e9, 2.5, 3.5]
    cval = np.array([9, 200, 31, 77], np.uint8)
    got = gpu.remap_maps(img, U, V, border=mode, border_value=cval, interpolation=interp)
    want = cpu_ref.remap(img, U, V, mode, cval, interpolation=interp)
    assert np.array_equal(got, want)


def test_tiny_sources_cubic(gpu):
    rng = np.random.default_rng(7)
    for sh, sw in ((1, 1), (2, 3), (3, 2), (4, 4)):
        img = rng.integers(0, 256, size=(sh, sw, 3), dtype=np.uint8)
        U = rng.uniform(-5, 8, size=(16, 16)).astype(np.float32)
        V = rng.uniform(-5, 8, size=(16, 16)).astype(np.float32)
        for mode in MODES:
            got = gpu.remap_maps(img, U, V, border=mode, interpolation=2)
            assert np.array_equal(got, cpu_ref.remap(img, U, V, mode, None, interpolation=2)), (sh, sw, mode)


def test_legacy_cli_files(gpu, legacy, synth, tmp_path):
    from PIL import Image

    src = tmp_path / "in"
    src.mkdir()
    pano = synth.synth_pano(512, 256, 2200, "S")
    Image.fromarray(pano).save(src / "room.png")
    (src / "notes.txt").write_text("not an image")
    out = tmp_path / "out"
    legacy.main(["--input_path", str(src), "--output_path", str(out), "--output_width", "100", "--output_height", "150",
                 "--pitch", "80", "--yaw_angles", "90", "0", "90", "--num_workers", "2"])
    names = sorted(p.name for p in out.iterdir())
    assert names == ["room_pitch80_yaw0_fov90.png", "room_pitch80_yaw90_fov90.png"]      # L:268
    got = np.asarray(Image.open(out / "room_pitch80_yaw90_fov90.png").convert("RGB"))
    U, V = maps.legacy_map(100, 150, float(np.radians(90)), float(np.radians(90)), float(np.radians(80)), 512, 256)
    want = cpu_ref.remap(pano, U, V, cpu_ref.BORDER_REFLECT)
    assert got.shape == (150, 100, 3)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1      # device map vs NumPy map: 1/32-px flips on a smooth image


def test_legacy_cli_exact_on_noise(gpu, legacy, synth, tmp_path):
    """The legacy tool with --exact: host-evaluated maps (the reference's own arithmetic, L:47-157), pixels on the GPU --
    the oracle's bytes on a NOISE panorama, every yaw; and back to the device maps afterwards."""
    from PIL import Image

    src = tmp_path / "in"
    src.mkdir()
    pano = synth.synth_pano(1024, 512, 2201, "N")
    Image.fromarray(pano).save(src / "n.png")
    out = tmp_path / "out"
    try:
        legacy.main(["--input_path", str(src), "--output_path", str(out), "--output_width", "200", "--output_height", "300",
                     "--pitch", "70", "--yaw_angles", "0", "60", "270", "--num_workers", "2", "--exact"])
        for yaw in (0, 60, 270):
            got = np.asarray(Image.open(out / f"n_pitch70_yaw{yaw}_fov90.png").convert("RGB"))
            U, V = maps.legacy_map(200, 300, float(np.radians(90)), float(np.radians(yaw)), float(np.radians(70)), 1024, 512)
            assert np.array_equal(got, cpu_ref.remap(pano, U, V, cpu_ref.BORDER_REFLECT)), yaw
    finally:
        legacy.set_exact(False)
    assert legacy._EXACT is False and legacy.precompute_mapping.cache_info().currsize == 0


@pytest.mark.parametrize("mode", MODES)
def test_resident_border_job_draws_the_oracles_bytes(gpu, synth, mode):
    """p2p_job_set_border: the legacy remap as a RESIDENT job (maps set once, several images through it) -- every border
    mode, three maps as the job's views, two images in turn; and the mode changed on a live job."""
    pw, ph, ow, oh = 1024, 512, 201, 150
    rng = np.random.default_rng(77 + mode)
    U = rng.uniform(-40.0, pw + 40.0, size=(3, oh, ow)).astype(np.float32)      # taps beyond every border
    V = rng.uniform(-30.0, ph + 30.0, size=(3, oh, ow)).astype(np.float32)
    panos = [synth.synth_pano(pw, ph, 2300 + i, "N") for i in range(2)]
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, pw, ph, 1, [0.0], [90.0] * 3, 90.0, ow, oh)
        job.set_border(mode)
        job.set_maps(None, U, V)
        for pano in panos:
            job.set_pano(0, pano)
            job.run()
            got = job.get_views(0)[0]
            for k in range(3):
                assert np.array_equal(got[k], cpu_ref.remap(pano, U[k], V[k], mode)), (mode, k)
        other = cpu_ref.BORDER_WRAP if mode != cpu_ref.BORDER_WRAP else cpu_ref.BORDER_REPLICATE
        job.set_border(other)
        job.run()
        assert np.array_equal(job.get_views(0)[0][1], cpu_ref.remap(panos[1], U[1], V[1], other))
        with pytest.raises(gpu.P2PError):
            job.set_border(9)
        job.close()
    finally:
        ctx.close()


def test_batched_maps_remap_equals_one_call_per_map(gpu, synth):
    """L:259-281: the legacy tool remaps one image through one precomputed map per yaw.  p2p_remap_maps_batch_u8
    draws them all in one launch; same bytes as cv2.remap per map (oracle), incl. a NaN coordinate under
    BORDER_REFLECT and an odd output size."""
    from oracle import cpu_ref, maps
    pano = synth.synth_pano(1024, 512, 4100, "N")
    UV = [maps.pitch_map_deg(200, 120, p, 1024, 512, 90) for p in (5, 60, 90, 150)]
    U, V = np.stack([u for u, _ in UV]), np.stack([v for _, v in UV])
    got = gpu.remap_maps_batch(pano, U, V, border=gpu.BORDER_REFLECT)
    for k in range(4):
        assert np.array_equal(got[k], cpu_ref.remap(pano, U[k], V[k], cpu_ref.BORDER_REFLECT)), k
        assert np.array_equal(got[k], gpu.remap_maps(pano, U[k], V[k], border=gpu.BORDER_REFLECT)), k
    U2, V2 = U[:, :33, :77].copy(), V[:, :33, :77].copy()
    got = gpu.remap_maps_batch(pano, U2, V2, border=gpu.BORDER_CONSTANT)
    for k in range(4):
        assert np.array_equal(got[k], cpu_ref.remap(pano, U2[k], V2[k], cpu_ref.BORDER_CONSTANT)), k
