"""CPU: the legacy tool's pieces (app/legacy/panorama_to_plane.py, "L").

* oracle/maps.py legacy_rotation_matrix / legacy_map against vectors produced by the reference's own
  get_rotation_matrix / precompute_mapping (tests/golden/legacy_maps_golden.npz, make_golden_legacy.py);
* the product's host-side mirror (get_rotation_matrix, check_pitch, check_yaw, CLI parser) against the same vectors;
* the oracle's INTER_NEAREST / INTER_CUBIC restatements against independent derivations."""
import argparse
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

from oracle import cpu_ref, maps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lg():
    z = np.load(os.path.join(ROOT, "tests", "golden", "legacy_maps_golden.npz"))
    return z, json.loads(bytes(z["meta_json"]).decode())


@pytest.fixture(scope="module")
def legacy(pkg):
    return importlib.import_module("360-to-planer-images_amd.panorama_to_plane")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_rotation_matrices_bit_exact(lg, legacy):
    z, meta = lg
    assert len(meta["rot"]) == 8
    for e in meta["rot"]:
        yr, pr = np.radians(e["yaw"]), np.radians(e["pitch"])
        want = z[e["key"]]
        # every entry of R_pitch @ R_yaw is 0, 1, one rounded cos/sin or ONE float32 product: identical on any host
        for R in (maps.legacy_rotation_matrix(yr, pr), legacy.get_rotation_matrix(yr, pr)):
            assert R.dtype == np.float32 and np.array_equal(R, want), e


def test_legacy_maps_tiny_and_cli_samples(lg, same_platform_as_golden):
    z, meta = lg
    for e in meta["tiny"]:
        U, V = maps.legacy_map(e["W"], e["H"], float(np.radians(e["fov"])), float(np.radians(e["yaw"])),
                               float(np.radians(e["pitch"])), e["pw"], e["ph"])
        gU, gV = z[e["key"] + "_U"], z[e["key"] + "_V"]
        ok = ~(np.isnan(gU) | np.isnan(gV) | np.isnan(U) | np.isnan(V))
        assert ok.mean() > 0.999
        dU = np.abs(U - gU)[ok]
        dU = np.minimum(dU, e["pw"] - 1 - dU)
        assert dU.max() <= 2e-3 and np.abs(V - gV)[ok].max() <= 2e-3   # float32 maps of a 256x128 panorama
        if same_platform_as_golden:
            assert np.array_equal(U, gU, equal_nan=True) and np.array_equal(V, gV, equal_nan=True)
    for e in meta["sampled"]:
        U, V = maps.legacy_map(e["W"], e["H"], float(np.radians(e["fov"])), float(np.radians(e["yaw"])),
                               float(np.radians(e["pitch"])), e["pw"], e["ph"])
        assert U.shape == (e["H"], e["W"]) and U.dtype == np.float32
        sU, sV = U[::e["stride_y"], ::e["stride_x"]], V[::e["stride_y"], ::e["stride_x"]]
        dU = np.abs(sU - z[e["key"] + "_U"])
        assert np.minimum(dU, e["pw"] - 1 - dU).max() <= 0.02 and np.abs(sV - z[e["key"] + "_V"]).max() <= 0.02
        if same_platform_as_golden:
            assert _sha(U) == e["sha_U"] and _sha(V) == e["sha_V"]


def test_check_yaw_and_check_pitch_like_the_reference(lg, legacy):
    _, meta = lg
    for e in meta["check_yaw"]:
        if "error" in e:
            with pytest.raises(argparse.ArgumentTypeError) as ex:
                legacy.check_yaw(list(e["in"]))
            assert str(ex.value) == e["error"]
        else:
            assert legacy.check_yaw(list(e["in"])) == e["out"]
    for e in meta["check_pitch"]:
        if "error" in e:
            with pytest.raises(argparse.ArgumentTypeError) as ex:
                legacy.check_pitch(e["in"])
            assert str(ex.value) == e["error"]
        else:
            assert legacy.check_pitch(e["in"]) == e["out"]


def test_legacy_cli_defaults(legacy):
    a = legacy.parse_arguments(["--input_path", "x"])   # L:285-301
    assert (a.output_path, a.output_format, a.FOV, a.output_width, a.output_height, a.pitch, a.num_workers) == \
           ("output_images", None, 90, 1000, 1500, 90, None)
    assert a.yaw_angles == [0, 60, 120, 180, 240, 300]
    a = legacy.parse_arguments(["--input_path", "x", "--yaw_angles", "300", "0", "0", "--pitch", "45", "--output_format", "jpg"])
    assert a.yaw_angles == [0, 300] and a.pitch == 45 and a.output_format == "jpg"
    with pytest.raises(SystemExit):
        legacy.parse_arguments(["--input_path", "x", "--pitch", "180"])
    with pytest.raises(argparse.ArgumentTypeError):
        legacy.parse_arguments(["--input_path", "x", "--yaw_angles", "361"])


# ---- INTER_NEAREST / INTER_CUBIC restatements ----
def test_cubic_weight_table_closed_form():
    t = cpu_ref.cubic_weight_table().astype(np.int64).reshape(32, 32, 4, 4)
    assert (t.reshape(1024, 16).sum(axis=1) == 32768).all()
    # separable up to rounding and the sum fix-up: |w[fy,fx,r,c] - 32768 * cy[r] * cx[c]| <= 1 except the fixed cell
    x = np.arange(32, dtype=np.float64) / 32
    A = -0.75
    c = np.stack([((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A, ((A + 2) * x - (A + 3)) * x * x + 1,
                  ((A + 2) * (1 - x) - (A + 3)) * (1 - x) ** 2 + 1], axis=1)
    c = np.concatenate([c, 1 - c.sum(axis=1, keepdims=True)], axis=1)          # (32, 4)
    ideal = 32768 * c[:, None, :, None] * c[None, :, None, :]
    err = np.abs(t - ideal)
    assert (err <= 0.51).mean() > 0.93 and err.max() <= 9       # the fix-up moves one entry per cell by < 9
    assert np.array_equal(t[0, 0].ravel(), [0, 0, 0, 0, 0, 32767, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0])
    assert np.array_equal(t[:, :, :, :], t.transpose(1, 0, 3, 2)) or (t != t.transpose(1, 0, 3, 2)).mean() < 0.01


def _py_cubic(img, U, V, border, cval):
    """Independent pure-Python INTER_CUBIC: exact rational arithmetic on the oracle's weight table."""
    from oracle.cpu_ref import quantise_maps
    t = cpu_ref.cubic_weight_table().astype(int)
    ix, iy, fx, fy = quantise_maps(U, V)
    h, w, cn = img.shape
    out = np.zeros(U.shape + (cn,), np.uint8)

    def bi(p, n):
        if 0 <= p < n:
            return p
        if border == 0:
            return -1
        if border == 1:
            return 0 if p < 0 else n - 1
        if border == 3:
            return p % n
        d = 1 if border == 4 else 0
        if n == 1:
            return 0
        while not 0 <= p < n:
            p = -p - 1 + d if p < 0 else n - 1 - (p - n) - d
        return p
    for y in range(U.shape[0]):
        for x in range(U.shape[1]):
            sx, sy = int(ix[y, x]) - 1, int(iy[y, x]) - 1
            wt = t[int(fy[y, x]) * 32 + int(fx[y, x])]
            if border == 0 and (sx >= w or sx + 4 <= 0 or sy >= h or sy + 4 <= 0):
                out[y, x] = cval[:cn]
                continue
            for k in range(cn):
                s = 0
                for r in range(4):
                    yy = bi(sy + r, h)
                    for c in range(4):
                        xx = bi(sx + c, w)
                        p = int(img[yy, xx, k]) if (yy >= 0 and xx >= 0) else int(cval[k])
                        s += p * int(wt[r * 4 + c])
                out[y, x, k] = min(255, max(0, (s + 16384) >> 15))
    return out


@pytest.mark.parametrize("border", [0, 1, 2, 3, 4])
def test_cubic_remap_against_python_derivation(border):
    rng = np.random.default_rng(50 + border)
    img = rng.integers(0, 256, size=(9, 11, 3), dtype=np.uint8)
    U = rng.uniform(-6, 17, size=(12, 14)).astype(np.float32)
    V = rng.uniform(-6, 15, size=(12, 14)).astype(np.float32)
    U[0, :3] = [np.nan, 4.0, 10.96875]
    V[0, :3] = [2.0, np.nan, 8.0]
    cval = np.array([7, 99, 250, 0], np.uint8)
    got = cpu_ref.remap(img, U, V, border, cval, interpolation=cpu_ref.INTER_CUBIC)
    assert np.array_equal(got, _py_cubic(img, U, V, border, cval))


def test_cubic_reproduces_pixels_at_integer_coordinates_and_overshoots_between():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(20, 20, 1), dtype=np.uint8)
    xs, ys = np.meshgrid(np.arange(2, 17, dtype=np.float32), np.arange(2, 17, dtype=np.float32))
    out = cpu_ref.remap(img, xs, ys, 0, None, interpolation=cpu_ref.INTER_CUBIC)
    assert np.array_equal(out, img[2:17, 2:17])       # the {32767, 1} cell still returns the centre pixel
    step = np.zeros((8, 8, 1), np.uint8)
    step[:, 4:] = 255
    o = cpu_ref.remap(step, np.full((1, 1), 2.5, np.float32), np.full((1, 1), 3.0, np.float32), 0, None,
                      interpolation=cpu_ref.INTER_CUBIC)
    assert o[0, 0, 0] == 0                              # negative lobe next to an edge, saturated at 0


@pytest.mark.parametrize("border", [0, 1, 2, 3, 4])
def test_nearest_remap(border):
    rng = np.random.default_rng(80 + border)
    img = rng.integers(0, 256, size=(7, 9, 3), dtype=np.uint8)
    U = rng.uniform(-12, 20, size=(10, 13)).astype(np.float32)
    V = rng.uniform(-10, 16, size=(10, 13)).astype(np.float32)
    U[0, :4] = [0.5, 1.5, 2.5, np.nan]     # half-even: 0, 2, 2; NaN -> -32768
    V[0, :4] = [0.0, 0.0, 0.0, 1.0]
    cval = np.array([1, 2, 3, 4], np.uint8)
    got = cpu_ref.remap(img, U, V, border, cval, interpolation=cpu_ref.INTER_NEAREST)
    sx = np.rint(np.nan_to_num(U, nan=-40000.0)).clip(-32768, 32767).astype(int)
    sy = np.rint(np.nan_to_num(V, nan=-40000.0)).clip(-32768, 32767).astype(int)
    h, w = img.shape[:2]
    for y in range(U.shape[0]):
        for x in range(U.shape[1]):
            a, b = sx[y, x], sy[y, x]
            if 0 <= a < w and 0 <= b < h:
                want = img[b, a]
            elif border == 0:
                want = cval[:3]
            elif border == 1:
                want = img[min(max(b, 0), h - 1), min(max(a, 0), w - 1)]
            elif border == 3:
                want = img[b % h, a % w]
            else:
                d = 1 if border == 4 else 0

                def refl(p, n):
                    while not 0 <= p < n:
                        p = -p - 1 + d if p < 0 else n - 1 - (p - n) - d
                    return p
                want = img[refl(b, h), refl(a, w)]
            assert np.array_equal(got[y, x], want), (x, y, a, b)
    assert np.array_equal(got[0, :3], img[0, [0, 2, 2]])
