"""The package's host-side pitch-map builder (360-to-planer-images_amd/_exact_maps.py, what --exact draws from) against the
vectors produced by the reference's own functions (tests/golden/maps_golden.npz, made by tests/golden/make_golden_maps.py
importing /root/reference/app/panorama_to_plane-pitch.py): bit for bit on the platform that made the fixtures -- the same
check oracle/maps.py passes -- and within 1e-5 elsewhere.  No GPU, nothing from oracle/ on the product's side."""
import hashlib
import importlib

import numpy as np
import pytest

from tests.conftest import PKG


@pytest.fixture(scope="module")
def em():
    return importlib.import_module(PKG + "._exact_maps")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _close(a, b, pw=None):
    ok = ~(np.isnan(a) | np.isnan(b))
    assert (np.isnan(a) == np.isnan(b)).mean() > 0.9999
    d = np.abs(a - b)
    if pw is not None:
        d = np.minimum(d, pw - 1 - d)
    return bool((d[ok] <= 1e-5 * np.maximum(np.abs(b[ok]), 1.0)).all())


def test_tiny_maps_are_the_references(em, golden, same_platform_as_golden):
    z, meta = golden
    n = 0
    for e in meta["tiny"]:
        if "pitch" not in e:
            continue
        U, V = em.pitch_mapping(e["ow"], e["oh"], np.radians(e["fov"]), np.radians(e["pitch"]), e["pw"], e["ph"])
        gU, gV = z[e["key"] + "_U"], z[e["key"] + "_V"]
        assert U.dtype == np.float32 and V.dtype == np.float32 and U.shape == gU.shape and U.flags["C_CONTIGUOUS"]
        assert _close(U, gU, e["pw"]) and _close(V, gV)
        if same_platform_as_golden:
            assert np.array_equal(U, gU, equal_nan=True) and np.array_equal(V, gV, equal_nan=True), e
        n += 1
    assert n == 18


def test_config_maps_are_the_references(em, golden, same_platform_as_golden):
    """cfg 1 / cfg 2 sized maps: stride samples everywhere, sha256 of the whole arrays on the golden platform."""
    z, meta = golden
    n = 0
    for e in meta["sampled"]:
        if e["ow"] == 4096 and e["pitch"] != 30:  # (config 4: one 16.8 Mpix map, the pole view)
            continue
        U, V = em.get_pitch_mapping(e["ow"], e["oh"], e["pitch"], e["pw"], e["ph"], e["fov"])
        st = e["stride"]
        assert _close(U[::st, ::st], z[e["key"] + "_U"], e["pw"]) and _close(V[::st, ::st], z[e["key"] + "_V"])
        if same_platform_as_golden:
            assert _sha(U) == e["sha_U"] and _sha(V) == e["sha_V"], e
        n += 1
    assert n >= 4
    em.clear()


def test_cache_keys_and_stack(em):
    """The reference's key (P:62) and one stack + one name per pitch list."""
    em.clear()
    a = em.get_pitch_mapping(64, 48, 60, 256, 128, 90)
    assert em.get_pitch_mapping(64, 48, 60, 256, 128, 90) is a
    assert (64, 48, 60, 256, 128, 90) in em.exact_pitch_mapping_cache
    U, V, k = em.pitch_map_stack(64, 48, [60, 90], 256, 128, 90)
    U2, V2, k2 = em.pitch_map_stack(64, 48, (60.0, 90.0), 256, 128, 90)
    assert U.shape == (2, 48, 64) and U.dtype == np.float32 and k != 0 and k2 == k and U2 is U
    assert np.array_equal(U[0], a[0]) and np.array_equal(V[0], a[1])
    _, _, k3 = em.pitch_map_stack(64, 48, [90, 60], 256, 128, 90)
    assert k3 != k
    em.clear()


def test_legacy_maps_are_the_references(em, same_platform_as_golden):
    """The legacy tool's --exact builder (_exact_maps.legacy_mapping, L:47-157) against the vectors the reference's own
    precompute_mapping produced (tests/golden/legacy_maps_golden.npz, make_golden_legacy.py)."""
    import json
    import os

    from tests.conftest import ROOT

    z = np.load(os.path.join(ROOT, "tests", "golden", "legacy_maps_golden.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    n = 0
    for e in meta["tiny"]:
        U, V = em.legacy_mapping(e["W"], e["H"], float(np.radians(e["fov"])), float(np.radians(e["yaw"])),
                                 float(np.radians(e["pitch"])), e["pw"], e["ph"])
        gU, gV = z[e["key"] + "_U"], z[e["key"] + "_V"]
        ok = ~(np.isnan(gU) | np.isnan(gV) | np.isnan(U) | np.isnan(V))
        dU = np.abs(U - gU)[ok]
        assert ok.mean() > 0.999 and np.minimum(dU, e["pw"] - 1 - dU).max() <= 2e-3 and np.abs(V - gV)[ok].max() <= 2e-3
        if same_platform_as_golden:
            assert np.array_equal(U, gU, equal_nan=True) and np.array_equal(V, gV, equal_nan=True), e
        n += 1
    for e in meta["sampled"]:
        U, V = em.legacy_mapping(e["W"], e["H"], float(np.radians(e["fov"])), float(np.radians(e["yaw"])),
                                 float(np.radians(e["pitch"])), e["pw"], e["ph"])
        if same_platform_as_golden:
            assert _sha(U) == e["sha_U"] and _sha(V) == e["sha_V"], e
        n += 1
    assert n >= 6


def test_random_geometries_equal_the_oracles_restatement(em):
    """Two independent restatements of P:114-175 / L:47-157 -- the package's builder (broadcast vectors, in-place steps) and
    oracle/maps.py (the reference's own statement order; pinned to the reference-made goldens) -- agree bit for bit on random
    geometries on whatever host this runs on: odd sizes, FOVs from 20 to 160 degrees, pitches next to the poles."""
    from oracle import maps
    rng = np.random.default_rng(606)
    for _ in range(60):
        W, H = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        pw = int(rng.integers(2, 5000))
        ph = max(1, pw // 2 + int(rng.integers(-3, 4)))
        fov, pitch, yaw = float(rng.uniform(20, 160)), float(rng.uniform(1, 179)), float(rng.uniform(-400, 400))
        a = em.pitch_mapping(W, H, np.radians(fov), np.radians(pitch), pw, ph)
        b = maps.pitch_map(W, H, np.radians(fov), np.radians(pitch), pw, ph)
        assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True), (W, H, pw, ph, fov, pitch)
        c = em.legacy_mapping(W, H, float(np.radians(fov)), float(np.radians(yaw)), float(np.radians(pitch)), pw, ph)
        d = maps.legacy_map(W, H, float(np.radians(fov)), float(np.radians(yaw)), float(np.radians(pitch)), pw, ph)
        assert np.array_equal(c[0], d[0], equal_nan=True) and np.array_equal(c[1], d[1], equal_nan=True), (W, H, pw, ph, fov, yaw, pitch)
