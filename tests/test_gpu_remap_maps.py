"""GPU parity for the legacy entry point panorama_to_plane(pano, U, V) == cv2.remap(..., INTER_LINEAR, border)
(L:159-194): bit-exact vs the CPU restatement for every border mode and channel count."""
import numpy as np
import pytest

from oracle import cpu_ref, maps

pytestmark = pytest.mark.gpu

MODES = [cpu_ref.BORDER_CONSTANT, cpu_ref.BORDER_REPLICATE, cpu_ref.BORDER_REFLECT, cpu_ref.BORDER_WRAP,
         cpu_ref.BORDER_REFLECT_101]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_random_maps_all_borders(gpu, mode, cn):
    rng = np.random.default_rng(100 + 10 * mode + cn)
    img = rng.integers(0, 256, size=(37, 53, cn), dtype=np.uint8)
    U = rng.uniform(-70, 130, size=(64, 80)).astype(np.float32)
    V = rng.uniform(-50, 90, size=(64, 80)).astype(np.float32)
    U[0, :6] = [np.nan, 0.0, 52.0, 52.5, -1.0, 1e9]
    V[0, :6] = [3.0, np.nan, 36.0, 36.5, -0.5, -1e9]
    cval = np.array([9, 200, 31, 77], np.uint8)
    got = gpu.remap_maps(img, U, V, border=mode, border_value=cval)
    want = cpu_ref.remap(img, U, V, mode, cval)
    assert np.array_equal(got, want)


def test_legacy_panorama_to_plane_entry_point(gpu, pkg, synth):
    pano = synth.synth_pano(1024, 512, 2000, "N")
    U, V = maps.pitch_map_deg(320, 200, 70, 1024, 512, 100)
    got = pkg.panorama_to_plane(pano, U, V)
    assert got.shape == (200, 320, 3) and got.dtype == np.uint8
    assert np.array_equal(got, cpu_ref.panorama_to_plane(pano, U, V))
    assert np.array_equal(pkg.interpolate_color(U, V, pano, "bilinear"), got)
    # an unknown method name falls back to bilinear, as the reference's dict.get(method, INTER_LINEAR) does (L:177)
    assert np.array_equal(pkg.interpolate_color(U, V, pano, "no such method"), got)


def test_legacy_path_equals_single_stage_of_current_path(gpu, pkg, synth):
    # yaw 0 is the identity resample, so the current tool's output equals one remap through the pitch map
    pano = synth.synth_pano(512, 256, 2001, "N")
    U, V = pkg.get_pitch_mapping(128, 96, 80, 512, 256, 90)
    one = pkg.panorama_to_plane(pano, U, V)
    two = pkg.process_yaw_and_pitchs(pano, 0, [80], 128, 96, 90)[0]
    assert np.array_equal(one, two)


def test_grayscale_2d_and_errors(gpu):
    img = np.arange(70, dtype=np.uint8).reshape(7, 10)
    U, V = np.meshgrid(np.arange(10, dtype=np.float32) + 0.25, np.arange(7, dtype=np.float32))
    got = gpu.remap_maps(img, U.astype(np.float32), V.astype(np.float32))
    assert got.shape == (7, 10) and np.array_equal(got, cpu_ref.remap(img, U, V))
    with pytest.raises(gpu.P2PError) as e:
        gpu.remap_maps(np.zeros((2, 32767, 1), np.uint8), U, V)  # cv::remap asserts cols < SHRT_MAX
    assert e.value.code == gpu.P2P_ERR_INVALID
    with pytest.raises(gpu.P2PError):
        gpu.remap_maps(img, U, V, border=7)


@pytest.mark.parametrize("mode", MODES)
def test_three_channel_maps_through_the_view_kernel(gpu, mode):
    # cn == 3 with a zero border value (or any non-constant border) is served by the fused view kernel with an
    # identity yaw stage: random maps incl. far out-of-range, NaN, exact edges; plus a smooth in-range map
    rng = np.random.default_rng(200 + mode)
    img = rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8)   # width divisible by 4: LDS path eligible
    U = rng.uniform(-40, 140, size=(70, 90)).astype(np.float32)
    V = rng.uniform(-30, 95, size=(70, 90)).astype(np.float32)
    U[0, :6] = [np.nan, 0.0, 95.0, 95.5, -1.0, 1e9]
    V[0, :6] = [3.0, np.nan, 63.0, 63.5, -0.5, -1e9]
    assert np.array_equal(gpu.remap_maps(img, U, V, border=mode), cpu_ref.remap(img, U, V, mode))
    yy, xx = np.mgrid[0:70, 0:90].astype(np.float32)
    Us, Vs = (3.0 + xx * 0.97 + yy * 0.05).astype(np.float32), (2.0 + yy * 0.8 + xx * 0.03).astype(np.float32)
    assert np.array_equal(gpu.remap_maps(img, Us, Vs, border=mode), cpu_ref.remap(img, Us, Vs, mode))


@pytest.mark.parametrize("mode", [cpu_ref.BORDER_CONSTANT, cpu_ref.BORDER_REFLECT])
def test_scattered_maps_on_a_large_image(gpu, synth, mode):
    """Random in-range coordinates on a big image: every tile's footprint rectangle is the whole image, so the
    view kernel's plan turns the tiles into sub-tiles with compacted item lists (or direct gathers) -- the
    result must still be cv2.remap's, byte for byte."""
    rng = np.random.default_rng(900 + mode)
    img = synth.synth_pano(2048, 1024, 2300, "N")
    U = rng.uniform(0.0, 2046.0, size=(128, 256)).astype(np.float32)
    V = rng.uniform(0.0, 1022.0, size=(128, 256)).astype(np.float32)
    # a smooth region in the middle (fits the LDS scheme) and a stripe that leaves the image (border taps)
    yy, xx = np.mgrid[0:48, 0:96].astype(np.float32)
    U[40:88, 80:176] = 700.0 + 1.3 * xx + 0.1 * yy
    V[40:88, 80:176] = 300.0 + 1.1 * yy
    U[100:104, :] += 1500.0
    got = gpu.remap_maps(img, U, V, border=mode)
    assert np.array_equal(got, cpu_ref.remap(img, U, V, mode))
    # the same maps as the pitch stage of the two-stage entry point, two yaws
    rows = np.stack([maps.yaw_column_table(2048, y) for y in (0, 77)])
    Uc, Vc = np.clip(U, 0, 2047), np.clip(V, 0, 1023)
    got2 = gpu.remap_views_maps(img, rows, Uc[None], Vc[None])
    from _util import oracle_views  # noqa: F401  (kept for symmetry with the other files)
    for yi, yaw in enumerate((0, 77)):
        rot = cpu_ref.yaw_stage(img, yaw)
        assert np.array_equal(got2[yi, 0], cpu_ref.remap(rot, Uc, Vc, cpu_ref.BORDER_CONSTANT))
