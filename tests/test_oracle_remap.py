"""Known-answer and cross-checks of the restated cv2.remap fixed-point gather (oracle/cv_remap_oracle.c).
The gather has no reference-held vectors (PARITY UNPINNED, see the C file's header); these tests pin
the restatement to the published OpenCV 4.10 algorithm through (a) the weight table's closed form,
(b) hand-computed answers, (c) an independent pure-Python re-derivation on small random cases."""
import numpy as np
import pytest

from oracle import cpu_ref


def test_weight_table_closed_form():
    wt = cpu_ref.weight_table().astype(np.int64)
    fx = np.arange(32)[None, :]
    fy = np.arange(32)[:, None]
    cf = np.stack([32 * (32 - fx) * (32 - fy), 32 * fx * (32 - fy), 32 * (32 - fx) * fy, 32 * fx * fy], -1).reshape(1024, 4)
    assert (wt[1:] == cf[1:]).all()
    assert wt[0].tolist() == [32767, 0, 0, 1]  # saturate_cast<short>(32768) + the sum fix-up
    assert (wt.sum(1) == 32768).all()


def test_quirk_cell_equals_closed_form_for_every_byte_pair():
    # fx = fy = 0: (32767*p00 + p11 + 16384) >> 15 == p00 for all bytes, so kernels may use 32*(32-fx)(32-fy)
    p00 = np.arange(256)[:, None]
    p11 = np.arange(256)[None, :]
    assert (((32767 * p00 + p11 + 16384) >> 15) == p00).all()


def test_shift_15_equals_shift_10_with_weights_over_32():
    rng = np.random.default_rng(0)
    p = rng.integers(0, 256, size=(20000, 4))
    fx = rng.integers(0, 32, size=20000)
    fy = rng.integers(0, 32, size=20000)
    w = np.stack([(32 - fx) * (32 - fy), fx * (32 - fy), (32 - fx) * fy, fx * fy], 1)
    a = ((32 * w * p).sum(1) + 16384) >> 15
    b = ((w * p).sum(1) + 512) >> 10
    assert (a == b).all()


def _ident_maps(h, w, dx=0.0, dy=0.0):
    U, V = np.meshgrid(np.arange(w, dtype=np.float32) + np.float32(dx), np.arange(h, dtype=np.float32) + np.float32(dy))
    return U.astype(np.float32), V.astype(np.float32)


def test_identity_and_integer_shift():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(17, 23, 3), dtype=np.uint8)
    U, V = _ident_maps(17, 23)
    assert np.array_equal(cpu_ref.remap(img, U, V), img)
    out = cpu_ref.remap(img, *_ident_maps(17, 23, dx=3, dy=2))
    assert np.array_equal(out[:15, :20], img[2:, 3:])
    assert (out[15:] == 0).all() and (out[:, 20:] == 0).all()  # BORDER_CONSTANT 0


def test_half_pixel_rounding():
    img = np.array([[[10, 20, 30], [13, 21, 255]]], dtype=np.uint8)
    out = cpu_ref.remap(img, np.array([[0.5]], np.float32), np.array([[0.0]], np.float32))
    assert out[0, 0].tolist() == [(10 + 13 + 1) >> 1, (20 + 21 + 1) >> 1, (30 + 255 + 1) >> 1]


def test_coordinate_quantisation_round_half_even_and_nan():
    U = np.array([[0.015625, 0.046875, 1.0 / 64 + 1e-4, np.nan, 3e9, -3e9]], np.float32)  # .5/32, 1.5/32 ties
    ix, iy, fx, fy = cpu_ref.quantise_maps(U, np.zeros_like(U))
    assert fx[0, :3].tolist() == [0, 2, 1]  # ties go to even
    assert ix[0, 3] == -32768 and ix[0, 4] == -32768 and ix[0, 5] == -32768  # cvtss2si "integer indefinite"
    img = np.full((4, 4, 3), 200, np.uint8)
    out = cpu_ref.remap(img, np.array([[np.nan, 1.0]], np.float32), np.array([[1.0, np.nan]], np.float32))
    assert (out == 0).all()


def _py_border(p, n, mode):
    if 0 <= p < n:
        return p
    if mode == cpu_ref.BORDER_CONSTANT:
        return -1
    if mode == cpu_ref.BORDER_REPLICATE:
        return 0 if p < 0 else n - 1
    if mode == cpu_ref.BORDER_WRAP:
        return p % n
    if n == 1:
        return 0
    period = 2 * n if mode == cpu_ref.BORDER_REFLECT else 2 * n - 2
    q = p % period
    if mode == cpu_ref.BORDER_REFLECT:
        return q if q < n else period - 1 - q
    return q if q < n else period - q


def _py_remap(img, U, V, mode, cval):
    """Independent derivation: exact rational bilinear on 1/32-quantised coordinates."""
    h, w, cn = img.shape
    out = np.zeros(U.shape + (cn,), np.uint8)
    for r in range(U.shape[0]):
        for c in range(U.shape[1]):
            u32, v32 = np.float32(U[r, c]) * np.float32(32), np.float32(V[r, c]) * np.float32(32)
            def q(v):
                if not (-2147483648.0 <= float(v) < 2147483648.0):
                    return -(2**31)
                return int(np.rint(v))
            sx, sy = q(u32), q(v32)
            ix, fx = max(-32768, min(32767, sx >> 5)), sx & 31
            iy, fy = max(-32768, min(32767, sy >> 5)), sy & 31
            if mode == cpu_ref.BORDER_CONSTANT and (ix >= w or ix + 1 < 0 or iy >= h or iy + 1 < 0):
                out[r, c] = cval[:cn]
                continue
            xs = [_py_border(ix, w, mode), _py_border(ix + 1, w, mode)]
            ys = [_py_border(iy, h, mode), _py_border(iy + 1, h, mode)]
            wx, wy = [32 - fx, fx], [32 - fy, fy]
            for k in range(cn):
                acc = 0
                for j in range(2):
                    for i in range(2):
                        p = int(img[ys[j], xs[i], k]) if xs[i] >= 0 and ys[j] >= 0 else int(cval[k])
                        acc += wx[i] * wy[j] * p
                out[r, c, k] = (acc + 512) >> 10
    return out


@pytest.mark.parametrize("mode", [cpu_ref.BORDER_CONSTANT, cpu_ref.BORDER_REPLICATE, cpu_ref.BORDER_REFLECT,
                                  cpu_ref.BORDER_WRAP, cpu_ref.BORDER_REFLECT_101])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_against_independent_python_derivation(mode, cn):
    rng = np.random.default_rng(10 * mode + cn)
    img = rng.integers(0, 256, size=(9, 13, cn), dtype=np.uint8)
    U = rng.uniform(-20, 33, size=(12, 15)).astype(np.float32)
    V = rng.uniform(-15, 24, size=(12, 15)).astype(np.float32)
    U[0, 0], V[0, 1] = np.nan, np.nan
    U[1, :4] = [0.0, 12.0, 12.5, -1.0]
    V[1, :4] = [8.0, 8.0, 8.96875, -0.03125]
    cval = np.array([7, 99, 250, 3], np.uint8)
    got = cpu_ref.remap(img, U, V, mode, cval)
    want = _py_remap(img, U, V, mode, cval)
    assert np.array_equal(got, want)


def test_single_channel_2d_input_and_size_assert():
    img = np.arange(35, dtype=np.uint8).reshape(5, 7)
    U, V = _ident_maps(5, 7)
    assert np.array_equal(cpu_ref.remap(img, U, V), img)
    with pytest.raises(ValueError):
        cpu_ref.remap(np.zeros((2, 32767, 1), np.uint8), U, V)  # cv::remap asserts cols < SHRT_MAX


def test_against_scipy_float_bilinear_within_quantisation():
    """Independent check of the geometric convention (integer coordinates are pixel centres, x = column,
    y = row, no half-pixel offset): scipy's float bilinear agrees with the fixed-point restatement to within
    the 1/32-px coordinate quantisation and the rounding, on a smooth image, for all three interpolations."""
    from scipy import ndimage

    yy, xx = np.mgrid[0:60, 0:80].astype(np.float64)
    img = (127 + 90 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 0.4 * xx).clip(0, 255).astype(np.uint8)
    rng = np.random.default_rng(11)
    U = rng.uniform(3, 75, size=(40, 50)).astype(np.float32)
    V = rng.uniform(3, 55, size=(40, 50)).astype(np.float32)
    ref = ndimage.map_coordinates(img.astype(np.float64), [V.astype(np.float64), U.astype(np.float64)], order=1)
    got = cpu_ref.remap(img, U, V, cpu_ref.BORDER_CONSTANT)
    # |d image / d coordinate| <= ~10 levels per pixel here, so 1/64 px of quantisation is < 0.2 level
    assert np.abs(got.astype(np.float64) - ref).max() <= 1.0
    near = cpu_ref.remap(img, U, V, cpu_ref.BORDER_CONSTANT, interpolation=cpu_ref.INTER_NEAREST)
    ref0 = ndimage.map_coordinates(img.astype(np.float64), [np.rint(V), np.rint(U)], order=0)
    assert np.array_equal(near.astype(np.float64), ref0)
    cub = cpu_ref.remap(img, U, V, cpu_ref.BORDER_CONSTANT, interpolation=cpu_ref.INTER_CUBIC)
    # cubic convolution (a = -0.75) on a smooth image stays close to the bilinear value
    assert np.abs(cub.astype(np.float64) - ref).max() <= 3.0
