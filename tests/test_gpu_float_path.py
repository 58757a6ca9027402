"""GPU: the opt-in float pixel path (P2P_FLAG_PIXELS_F32 / _F16; BASELINE config 5's "fp16 pixel path ...
tolerance check vs fp32 reference").  The reference has no such path, so the checks are:
  * float32 kernel vs a plain NumPy float32 evaluation of the same one-resample formula: <= 1 level
    (the device map differs from NumPy's in the last bit for some pixels);
  * float16 kernel vs the float32 kernel: <= 1 level (written tolerance of config 5);
  * both vs the exact two-stage uint8 path on band-limited panoramas: <= 2 levels away from the seam column,
    where the exact path deliberately reproduces the reference's clipped (non-wrapping) yaw map."""
import numpy as np
import pytest

from oracle import maps

pytestmark = pytest.mark.gpu


def numpy_float_views(pano, yaws, pitches, ow, oh, fov=90):
    """out[yaw][pitch] = bilinear(pano, U_pitch + yaw * pw / 360 (mod pw, wrap-around), V_pitch), float32."""
    ph, pw = pano.shape[:2]
    P = pano.astype(np.float32)
    out = np.zeros((len(yaws), len(pitches), oh, ow, 3), np.uint8)
    for pi, pitch in enumerate(pitches):
        U, V = maps.pitch_map_deg(ow, oh, pitch, pw, ph, fov, clip_u=False)  # the float path wraps, it does not clip
        dead = np.isnan(U) | np.isnan(V)
        U, V = np.nan_to_num(U), np.nan_to_num(V)
        y0 = V.astype(np.int32)
        wy = (V - y0.astype(np.float32))[..., None]
        y1 = np.minimum(y0 + 1, ph - 1)
        for yi, yaw in enumerate(yaws):
            sh = np.float32(np.fmod(np.radians(yaw) * pw / (2 * np.pi), pw) % pw)
            xs = U + sh
            xs = np.where(xs >= np.float32(pw), xs - np.float32(pw), xs)
            x0 = np.minimum(xs.astype(np.int32), pw - 1)
            wx = (xs - x0.astype(np.float32))[..., None]
            x1 = np.where(x0 + 1 < pw, x0 + 1, 0)
            a, b, c, d = P[y0, x0], P[y0, x1], P[y1, x0], P[y1, x1]
            h0 = wx * (b - a) + a
            h1 = wx * (d - c) + c
            v = np.rint(wy * (h1 - h0) + h0).clip(0, 255).astype(np.uint8)
            v[dead] = 0
            out[yi, pi] = v
    return out


CFGS = [
    dict(pw=2048, ph=1024, ow=512, oh=512, yaws=[0], pitches=[90], fov=90),                       # config 1
    dict(pw=4096, ph=2048, ow=640, oh=360, yaws=[0, 1, 30, 123, 359], pitches=[60, 90, 120], fov=90),
    dict(pw=2048, ph=1024, ow=333, oh=250, yaws=[77, 200], pitches=[20, 150], fov=100),            # poles, odd sizes
]


@pytest.mark.parametrize("cfg", CFGS)
def test_float32_path_matches_numpy_float32(gpu, synth, cfg):
    pano = synth.synth_pano(cfg["pw"], cfg["ph"], 5000, "S")
    got = gpu.remap_views(pano, cfg["yaws"], cfg["pitches"], cfg["fov"], cfg["ow"], cfg["oh"], flags=gpu.FLAG_PIXELS_F32)
    want = numpy_float_views(pano, cfg["yaws"], cfg["pitches"], cfg["ow"], cfg["oh"], cfg["fov"])
    d = np.abs(got.astype(int) - want.astype(int))
    print("f32 vs numpy: max %d, differing %.4f" % (d.max(), (d > 0).mean()))
    assert d.max() <= 1 and (d > 0).mean() < 0.02


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("kind", ["S", "N"])
def test_float16_path_within_one_level_of_float32(gpu, synth, cfg, kind):
    pano = synth.synth_pano(cfg["pw"], cfg["ph"], 5001, kind)
    a = gpu.remap_views(pano, cfg["yaws"], cfg["pitches"], cfg["fov"], cfg["ow"], cfg["oh"], flags=gpu.FLAG_PIXELS_F32)
    b = gpu.remap_views(pano, cfg["yaws"], cfg["pitches"], cfg["fov"], cfg["ow"], cfg["oh"], flags=gpu.FLAG_PIXELS_F16)
    d = np.abs(a.astype(int) - b.astype(int))
    print("f16 vs f32 (%s): max %d, differing %.4f" % (kind, d.max(), (d > 0).mean()))
    assert d.max() <= 1


def test_config5_style_sweep_f16_vs_f32_and_vs_exact(gpu, synth):
    # config 5 at reduced size: 1-degree yaw sweep, one pitch
    pw, ph, ow, oh = 4096, 2048, 480, 270
    pano = synth.synth_pano(pw, ph, 5002, "S")
    yaws = list(range(0, 360, 3))
    f32 = gpu.remap_views(pano, yaws, [90], 90, ow, oh, flags=gpu.FLAG_PIXELS_F32)
    f16 = gpu.remap_views(pano, yaws, [90], 90, ow, oh, flags=gpu.FLAG_PIXELS_F16)
    exact = gpu.remap_views(pano, yaws, [90], 90, ow, oh)
    assert np.abs(f32.astype(int) - f16.astype(int)).max() <= 1
    # the exact path clips at the seam column (no wrap) as the reference does; compare away from it
    U, _ = maps.pitch_map_deg(ow, oh, 90, pw, ph, 90)
    worst = 0
    for yi, yaw in enumerate(yaws):
        src_col = (U + yaw * pw / 360.0) % pw
        away = (src_col > 2) & (src_col < pw - 3)
        d = np.abs(f32[yi, 0].astype(int) - exact[yi, 0].astype(int)).max(axis=-1)
        worst = max(worst, int(d[away].max()))
    print("f32 vs exact two-stage path away from the seam: max", worst)
    assert worst <= 2


def test_float_path_wraps_at_the_seam_where_the_exact_path_clips(gpu):
    # a panorama that is black except its first column: only a wrapping resample blends it into column pw-1+f
    pw, ph = 512, 256
    pano = np.zeros((ph, pw, 3), np.uint8)
    pano[:, 0] = 255
    exact = gpu.remap_views(pano, [0], [90], 90, 128, 96)
    wrap = gpu.remap_views(pano, [0], [90], 90, 128, 96, flags=gpu.FLAG_PIXELS_F32)
    U, _ = maps.pitch_map_deg(128, 96, 90, pw, ph, 90)
    seam = (U > pw - 1)          # coordinates between the last and the first column
    assert seam.sum() == 0 or (wrap[0, 0][seam] > 0).any()
    assert wrap.shape == exact.shape


def test_float_path_blends_across_the_seam_of_the_map_itself(gpu):
    """The pitch map's own azimuth in (pw - 1, pw) -- which P:172 clips onto column pw - 1 -- is resampled between
    column pw - 1 and column 0 by the float path (it used to inherit the clip and repeat a texel)."""
    pw, ph, ow, oh, pitch, fov = 256, 128, 320, 200, 25, 120   # the pole is in view: every azimuth occurs
    pano = np.zeros((ph, pw, 3), np.uint8)
    pano[:, 0] = 200     # first column bright, last column dark: a blend shows up as intermediate values
    pano[:, pw - 1] = 40
    U, V = maps.pitch_map_deg(ow, oh, pitch, pw, ph, fov, clip_u=False)
    band = (U > pw - 1) & (U < pw) & ~np.isnan(V)
    assert band.sum() > 0
    got = gpu.remap_views(pano, [0], [pitch], fov, ow, oh, flags=gpu.FLAG_PIXELS_F32)[0, 0]
    want = numpy_float_views(pano, [0], [pitch], ow, oh, fov)[0, 0]
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    vals = got[band][:, 0]
    assert ((vals > 45) & (vals < 195)).any()  # genuinely between the two columns


def test_float_path_rejects_caller_maps(gpu, synth):
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, 256, 128, 1, [0], [90], 90, 64, 48, flags=gpu.FLAG_PIXELS_F16)
    job.set_pano(0, synth.synth_pano(256, 128, 1, "N"))
    job.set_maps(None, np.zeros((1, 48, 64), np.float32), np.zeros((1, 48, 64), np.float32))
    with pytest.raises(gpu.P2PError) as e:
        job.run()
    assert e.value.code == gpu.P2P_ERR_STATE
    job.close()
    ctx.close()


def test_pixel_centre_convention(gpu, synth):
    """P2P_FLAG_PIXEL_CENTRES (float paths only): rays through output-pixel centres, texel i centred at i + 0.5."""
    pw, ph, ow, oh, fov, pitch, yaw = 1024, 512, 200, 120, 90, 75, 33
    pano = synth.synth_pano(pw, ph, 5100, "S")
    got = gpu.remap_views(pano, [yaw], [pitch], fov, ow, oh, flags=gpu.FLAG_PIXELS_F32 | gpu.FLAG_PIXEL_CENTRES)
    # NumPy float32 evaluation of the same formula (P:114-175 with u + 0.5, v + 0.5)
    f = np.float32((0.5 * ow) / np.tan(np.radians(fov) / 2))
    u, v = np.meshgrid(np.arange(ow, dtype=np.float32) + np.float32(0.5), np.arange(oh, dtype=np.float32) + np.float32(0.5))
    x, y, z = u - np.float32(ow / 2.0), np.float32(oh / 2.0) - v, np.full((oh, ow), f, np.float32)
    n = np.sqrt(x * x + y * y + z * z)
    x, y, z = x / n, y / n, z / n
    c, s = np.float32(np.cos(np.radians(pitch))), np.float32(np.sin(np.radians(pitch)))
    yr, zr = c * y - s * z, s * y + c * z
    U = (np.arctan2(yr, x) % np.float32(2 * np.pi)) * np.float32(pw) / np.float32(2 * np.pi)
    V = np.arccos(zr) * np.float32(ph) / np.float32(np.pi)
    V = np.clip(V, 0, ph - 1)  # the float path wraps U instead of clipping it
    xs = (U + np.float32(yaw * pw / 360.0) - np.float32(0.5)) % np.float32(pw)
    ys = np.maximum(V - np.float32(0.5), 0)
    x0 = np.minimum(xs.astype(np.int32), pw - 1); wx = (xs - x0)[..., None]
    x1 = (x0 + 1) % pw
    y0 = ys.astype(np.int32); wy = (ys - y0)[..., None]; y1 = np.minimum(y0 + 1, ph - 1)
    P = pano.astype(np.float32)
    h0 = P[y0, x0] + wx * (P[y0, x1] - P[y0, x0]); h1 = P[y1, x0] + wx * (P[y1, x1] - P[y1, x0])
    want = np.rint(h0 + wy * (h1 - h0)).clip(0, 255).astype(np.uint8)
    assert np.abs(got[0, 0].astype(int) - want.astype(int)).max() <= 1
    plain = gpu.remap_views(pano, [yaw], [pitch], fov, ow, oh, flags=gpu.FLAG_PIXELS_F32)
    assert (plain != got).mean() > 0.2                       # it is a different picture (half a pixel apart)
    with pytest.raises(gpu.P2PError) as e:
        gpu.remap_views(pano, [yaw], [pitch], fov, ow, oh, flags=gpu.FLAG_PIXEL_CENTRES)
    assert e.value.code == gpu.P2P_ERR_INVALID
