"""Shared helpers for the parity tests (the oracle is the checker, never the thing under test)."""
import numpy as np

from oracle import cpu_ref, maps


def oracle_views(pano, yaws, pitches, ow, oh, fov=90):
    """[n_yaw][n_pitch][oh][ow][3] from the CPU restatement of P:181-221."""
    cache = {}
    out = np.empty((len(yaws), len(pitches), oh, ow, 3), dtype=np.uint8)
    for yi, y in enumerate(yaws):
        sl = cpu_ref.process_yaw_and_pitchs(pano, y, pitches, ow, oh, fov, _pitch_cache=cache)
        for pi in range(len(pitches)):
            out[yi, pi] = sl[pi]
    return out


def oracle_views_threaded(pano, yaws, pitches, ow, oh, fov=90, threads=None):
    """oracle_views with one task per yaw on a thread pool, the reference's own parallel unit (P:252-265); the C
    restatement releases the GIL.  For the full-size configurations, where one thread would take minutes."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    ph, pw = pano.shape[:2]
    cache = {(ow, oh, p, pw, ph, fov): maps.pitch_map_deg(ow, oh, p, pw, ph, fov) for p in pitches}
    out = np.empty((len(yaws), len(pitches), oh, ow, 3), dtype=np.uint8)

    def one(yi):
        sl = cpu_ref.process_yaw_and_pitchs(pano, yaws[yi], pitches, ow, oh, fov, _pitch_cache=cache)
        for pi in range(len(pitches)):
            out[yi, pi] = sl[pi]

    n = threads or max(1, min(len(yaws), int((os.cpu_count() or 1) * 0.9)))
    with ThreadPoolExecutor(max_workers=n) as ex:
        list(ex.map(one, range(len(yaws))))
    return out


def oracle_maps(yaws, pitches, ow, oh, pw, ph, fov=90):
    rows = np.stack([maps.yaw_column_table(pw, y) for y in yaws])
    UV = [maps.pitch_map_deg(ow, oh, p, pw, ph, fov) for p in pitches]
    return rows, np.stack([u for u, _ in UV]), np.stack([v for _, v in UV])


def coords_to_maps(coords):
    """Quantised (sx, sy) in 1/32 px (INT32_MIN = NaN) -> float32 maps that re-quantise to the same values."""
    sx = coords[..., 0].astype(np.float64)
    sy = coords[..., 1].astype(np.float64)
    U = (sx / 32.0).astype(np.float32)
    V = (sy / 32.0).astype(np.float32)
    nan = (coords[..., 0] == np.iinfo(np.int32).min) | (coords[..., 1] == np.iinfo(np.int32).min)
    U[nan] = np.nan
    V[nan] = np.nan
    return U, V


def diff_stats(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return int(d.max()), float((d > 1).mean()), float((d > 0).mean())


_hip = None


def poison_views(job, byte=0xA5):
    """Fill a job's device output block with a byte pattern (straight through the HIP runtime: a test's business, no
    knob of the library): the block comes out of the library's device memory pool, where an earlier job of the same
    size may have left the very views the test expects -- a tile that nobody draws would otherwise go unnoticed."""
    import ctypes

    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        _hip.hipMemset.restype = ctypes.c_int
        _hip.hipDeviceSynchronize.restype = ctypes.c_int
    ptr, n = job.device_out()
    assert ptr and n > 0
    assert _hip.hipMemset(ctypes.c_void_p(ptr), int(byte), n) == 0 and _hip.hipDeviceSynchronize() == 0
