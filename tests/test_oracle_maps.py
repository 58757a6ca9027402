"""The oracle's map builders (oracle/maps.py) against vectors produced by the reference's own
functions (tests/golden/maps_golden.npz, generator: tests/golden/make_golden_maps.py)."""
import hashlib

import numpy as np

from oracle import maps


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _close(a, b, pw=None):
    ok = ~(np.isnan(a) | np.isnan(b))
    assert (np.isnan(a) == np.isnan(b)).mean() > 0.9999
    d = np.abs(a - b)
    if pw is not None:  # azimuth seam: 0 and pw-1 are neighbours
        d = np.minimum(d, pw - 1 - d)
    return bool((d[ok] <= 1e-5 * np.maximum(np.abs(b[ok]), 1.0)).all())


def test_tiny_pitch_maps_full(golden, same_platform_as_golden):
    z, meta = golden
    n = 0
    for e in meta["tiny"]:
        if "pitch" not in e:
            continue
        U, V = maps.pitch_map_deg(e["ow"], e["oh"], e["pitch"], e["pw"], e["ph"], e["fov"])
        gU, gV = z[e["key"] + "_U"], z[e["key"] + "_V"]
        assert U.dtype == np.float32 and U.shape == gU.shape
        assert _close(U, gU, e["pw"]) and _close(V, gV)
        if same_platform_as_golden:
            assert np.array_equal(U, gU, equal_nan=True) and np.array_equal(V, gV, equal_nan=True)
        n += 1
    assert n == 18


def test_tiny_yaw_maps_full_bit_exact(golden):
    z, meta = golden
    for e in meta["tiny"]:
        if "yaw" not in e:
            continue
        U, V = maps.yaw_map(e["pw"], e["ph"], e["yaw"])
        # IEEE-only arithmetic (float32 mul/div, float64 add/fmod/mul/div): identical on every host
        assert np.array_equal(U, z[e["key"] + "_U"]) and np.array_equal(V, z[e["key"] + "_V"])


def test_yaw_column_tables_bit_exact(golden):
    z, meta = golden
    assert len(meta["yaw_tables"]) == 30
    for e in meta["yaw_tables"]:
        assert np.array_equal(maps.yaw_column_table(e["pw"], e["yaw"]), z[e["key"]]), e


def test_sampled_config_maps(golden, same_platform_as_golden):
    z, meta = golden
    for e in meta["sampled"]:
        if e["ow"] == 4096:  # config 4 maps: 16.8 Mpix each, several seconds; covered by one entry below
            continue
        U, V = maps.pitch_map_deg(e["ow"], e["oh"], e["pitch"], e["pw"], e["ph"], e["fov"])
        st = e["stride"]
        assert _close(U[::st, ::st], z[e["key"] + "_U"], e["pw"])
        assert _close(V[::st, ::st], z[e["key"] + "_V"])
        if same_platform_as_golden:
            assert _sha(U) == e["sha_U"] and _sha(V) == e["sha_V"]


def test_config4_pole_map_sample(golden, same_platform_as_golden):
    z, meta = golden
    e = [m for m in meta["sampled"] if m["ow"] == 4096 and m["pitch"] == 30][0]
    U, V = maps.pitch_map_deg(e["ow"], e["oh"], e["pitch"], e["pw"], e["ph"], e["fov"])
    st = e["stride"]
    assert _close(U[::st, ::st], z[e["key"] + "_U"], e["pw"]) and _close(V[::st, ::st], z[e["key"] + "_V"])
    if same_platform_as_golden:
        assert _sha(U) == e["sha_U"] and _sha(V) == e["sha_V"]
    # the pole is inside this view: azimuth covers the whole panorama width (seam)
    assert U.min() == 0.0 and U.max() == e["pw"] - 1


def test_nan_pixels(golden, same_platform_as_golden):
    _, meta = golden
    for e in meta["nan_pixels"]:
        if e["ow"] != 1920:
            continue
        U, V = maps.pitch_map_deg(e["ow"], e["oh"], e["pitch"], e["pw"], e["ph"], e["fov"])
        assert int(np.isnan(U).sum()) == e["nan_U"] == 0
        if same_platform_as_golden:  # which pixel rounds above 1.0 depends on the host's sgemm/arccos
            assert np.argwhere(np.isnan(V)).tolist() == e["nan_V"]
        assert np.isnan(V).sum() <= 2


def test_geometry_known_answers(golden):
    _, meta = golden
    known = {k["what"]: k for k in meta["known"]}
    # view centre = (3*pw/4, ph*pitch/180): SURVEY 3.4
    assert known["centre_512_p90"]["U"] == 1536.0 and known["centre_512_p90"]["V"] == 512.0
    U, V = maps.pitch_map_deg(512, 512, 90, 2048, 1024, 90)
    assert U[256, 256] == 1536.0 and V[256, 256] == 512.0
    U, V = maps.pitch_map_deg(1920, 1080, 60, 8192, 4096, 90)
    assert abs(U[540, 960] - known["centre_1080p_p60"]["U"]) < 1e-2 and abs(V[540, 960] - 4096 * 60 / 180) < 1e-2
    # yaw 30 at 2048: column 1877 is clamped to pw-1 (no wrap interpolation), 1878 restarts at 0.667
    row = maps.yaw_column_table(2048, 30)
    assert [float(x) for x in row[1876:1880]] == known["yaw30_2048_cols_1876_1879"]["U"]
    assert row[1877] == 2047.0 and 0.6 < row[1878] < 0.7


def test_yaw_zero_is_identity_and_integer_shift_is_exact():
    from oracle import cpu_ref

    # float32 phi carries ~6e-8 relative error, so U is x only to ~1e-3; its 1/32-px quantisation is exact
    row = maps.yaw_column_table(4096, 0)
    assert np.abs(row - np.arange(4096)).max() < 1e-3
    ix, _, fx, _ = cpu_ref.quantise_maps(row[None], np.zeros((1, 4096), np.float32))
    assert np.array_equal(ix[0], np.arange(4096)) and (fx == 0).all()
    row = maps.yaw_column_table(8192, 45)  # 45 deg * 8192 / 360 = 1024 columns
    ix, _, fx, _ = cpu_ref.quantise_maps(row[None], np.zeros((1, 8192), np.float32))
    assert np.array_equal(ix[0], (np.arange(8192) + 1024) % 8192) and (fx == 0).all()
