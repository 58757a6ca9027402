"""ctypes binding of libp2p_hip.so (the C ABI declared in include/p2p_hip.h).

There is NO CPU fallback: if the shared library is missing this module raises at
load time, and if no HIP device is usable every compute call raises P2PError
(P2P_ERR_NO_DEVICE).  Build the library with `python -c "import __graft_entry__ as g; g.build()"`
or `python 360-to-planer-images_amd/_build.py`.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# P2P_LIB_PATH: load another build of the same library (A/B timing of kernel variants, diagnostic builds)
LIB_PATH = os.environ.get("P2P_LIB_PATH") or os.path.join(_HERE, "libp2p_hip.so")

P2P_OK = 0
P2P_ERR_INVALID, P2P_ERR_NO_DEVICE, P2P_ERR_HIP, P2P_ERR_OOM, P2P_ERR_STATE = -1, -2, -3, -4, -5
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC = 0, 1, 2
FLAG_PIXELS_F32 = 4   # opt-in float pixel path (beyond the reference): one float32 resample per view
FLAG_PIXELS_F16 = 8   # the same with the 2x2 blend in packed float16
FLAG_PIXEL_CENTRES = 16  # float paths only: sample through pixel centres (not the reference's convention)

# every symbol include/p2p_hip.h declares (tests check the library exports exactly these)
ABI_SYMBOLS = (
    "p2p_version", "p2p_last_error", "p2p_device_count",
    "p2p_remap_views_u8", "p2p_remap_views_f64", "p2p_remap_views_maps_u8", "p2p_remap_views_pitch_maps_f64", "p2p_remap_maps_u8", "p2p_remap_maps_interp_u8",
    "p2p_remap_maps_batch_u8",
    "p2p_build_pitch_map", "p2p_build_yaw_row", "p2p_build_rot_map",
    "p2p_ctx_create", "p2p_ctx_destroy", "p2p_ctx_synchronize", "p2p_ctx_mark", "p2p_ctx_marked_ms",
    "p2p_job_time_launches", "p2p_job_plan_ms",
    "p2p_job_create", "p2p_job_create_f64", "p2p_job_set_yaws_f64", "p2p_job_destroy", "p2p_job_set_pano",
    "p2p_job_set_pano_async", "p2p_job_share_panos", "p2p_job_get_views_async", "p2p_job_wait", "p2p_job_set_maps", "p2p_job_run",
    "p2p_job_get_views", "p2p_job_kernel_ms", "p2p_job_kernel_ms_last", "p2p_job_device_out", "p2p_job_get_coords",
    "p2p_job_get_yaw_tables", "p2p_job_set_yaws", "p2p_host_alloc", "p2p_host_free", "p2p_release_cache",
    "p2p_reload_options", "p2p_job_get_info", "p2p_job_get_view", "p2p_job_get_view_async", "p2p_job_set_view_mask",
    "p2p_device_mem_info", "p2p_job_set_border", "p2p_job_set_rows", "p2p_job_get_view_rows", "p2p_job_get_view_rows_async",
)


class P2PError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libp2p_hip: %s (status %d)" % (message, code))
        self.code = code


class JobInfo(ctypes.Structure):
    """p2p_job_info (include/p2p_hip.h): how a job is drawn -- for tests and tools."""
    _fields_ = [
        ("tile_w", ctypes.c_int32), ("tile_h", ctypes.c_int32), ("pairs_per_block", ctypes.c_int32),
        ("pair_chunks", ctypes.c_int32), ("list_order", ctypes.c_int32), ("main_group", ctypes.c_int32),
        ("prefetch_lead", ctypes.c_int32), ("n_odd_yaws", ctypes.c_int32), ("n_tiles", ctypes.c_int64),
        ("n_gather_tiles", ctypes.c_int64), ("timing_events", ctypes.c_int32), ("copy_streams", ctypes.c_int32),
        ("n_views_wanted", ctypes.c_int32), ("chunks_per_workgroup", ctypes.c_int32), ("band_tiles", ctypes.c_int32), ("lds_items_cap", ctypes.c_int32),
    ]


class JobDesc(ctypes.Structure):
    _fields_ = [
        ("pw", ctypes.c_int32), ("ph", ctypes.c_int32), ("n_panos", ctypes.c_int32),
        ("n_yaw", ctypes.c_int32), ("yaw_deg", ctypes.POINTER(ctypes.c_int32)),
        ("n_pitch", ctypes.c_int32), ("pitch_deg", ctypes.POINTER(ctypes.c_int32)),
        ("fov_deg", ctypes.c_int32), ("ow", ctypes.c_int32), ("oh", ctypes.c_int32),
        ("flags", ctypes.c_int32),
    ]


_lib = None


def lib():
    """Load libp2p_hip.so once.  Raises OSError (with build instructions) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            "%s is missing: the HIP extension has not been built and there is no CPU fallback. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` in the repository root." % LIB_PATH
        )
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_i64, c_vp, c_dbl = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
    L.p2p_version.restype = ctypes.c_char_p
    L.p2p_version.argtypes = []
    L.p2p_last_error.restype = ctypes.c_char_p
    L.p2p_last_error.argtypes = []
    L.p2p_device_count.restype = c_int
    L.p2p_device_count.argtypes = []
    L.p2p_remap_views_u8.restype = c_int
    L.p2p_remap_views_u8.argtypes = [c_vp, c_int, c_int, c_i64, c_vp, c_int, c_vp, c_int, c_int,
                                     c_int, c_int, c_vp, c_int, c_int]
    L.p2p_remap_views_f64.restype = c_int
    L.p2p_remap_views_f64.argtypes = [c_vp, c_int, c_int, c_i64, c_vp, c_int, c_vp, c_int, c_dbl,
                                      c_int, c_int, c_vp, c_int, c_int]
    L.p2p_remap_views_maps_u8.restype = c_int
    L.p2p_remap_views_maps_u8.argtypes = [c_vp, c_int, c_int, c_i64, c_vp, c_int, c_vp, c_vp, c_int,
                                          c_int, c_int, c_vp, c_int]
    L.p2p_remap_views_pitch_maps_f64.restype = c_int
    L.p2p_remap_views_pitch_maps_f64.argtypes = [c_vp, c_int, c_int, c_i64, c_vp, c_int, c_vp, c_vp, c_int, ctypes.c_uint64,
                                                 c_int, c_int, c_vp, c_int]
    L.p2p_remap_maps_u8.restype = c_int
    L.p2p_remap_maps_u8.argtypes = [c_vp, c_int, c_int, c_i64, c_int, c_vp, c_vp, c_int, c_int, c_vp,
                                    c_int, c_vp, c_int]
    L.p2p_remap_maps_batch_u8.restype = c_int
    L.p2p_remap_maps_batch_u8.argtypes = [c_vp, c_int, c_int, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int]
    L.p2p_remap_maps_interp_u8.restype = c_int
    L.p2p_remap_maps_interp_u8.argtypes = [c_vp, c_int, c_int, c_i64, c_int, c_vp, c_vp, c_int, c_int, c_vp,
                                           c_int, c_int, c_vp, c_int]
    L.p2p_build_pitch_map.restype = c_int
    L.p2p_build_pitch_map.argtypes = [c_int, c_int, c_dbl, c_dbl, c_int, c_int, c_vp, c_vp, c_int]
    L.p2p_build_rot_map.restype = c_int
    L.p2p_build_rot_map.argtypes = [c_int, c_int, c_dbl, c_vp, c_int, c_int, c_vp, c_vp, c_int]
    L.p2p_build_yaw_row.restype = c_int
    L.p2p_build_yaw_row.argtypes = [c_int, c_dbl, c_vp, c_int]
    L.p2p_ctx_create.restype = c_int
    L.p2p_ctx_create.argtypes = [c_int, ctypes.POINTER(c_vp)]
    L.p2p_ctx_destroy.restype = None
    L.p2p_ctx_destroy.argtypes = [c_vp]
    L.p2p_ctx_synchronize.restype = c_int
    L.p2p_ctx_synchronize.argtypes = [c_vp]
    L.p2p_ctx_mark.restype = c_int
    L.p2p_ctx_mark.argtypes = [c_vp, c_int]
    L.p2p_ctx_marked_ms.restype = c_int
    L.p2p_ctx_marked_ms.argtypes = [c_vp, ctypes.POINTER(ctypes.c_float)]
    L.p2p_job_time_launches.restype = c_int
    L.p2p_job_time_launches.argtypes = [c_vp, c_int]
    L.p2p_job_create.restype = c_int
    L.p2p_job_create.argtypes = [c_vp, ctypes.POINTER(JobDesc), ctypes.POINTER(c_vp)]
    L.p2p_job_create_f64.restype = c_int
    L.p2p_job_create_f64.argtypes = [c_vp, ctypes.POINTER(JobDescF64), ctypes.POINTER(c_vp)]
    L.p2p_job_set_yaws_f64.restype = c_int
    L.p2p_job_set_yaws_f64.argtypes = [c_vp, c_vp]
    L.p2p_job_destroy.restype = None
    L.p2p_job_destroy.argtypes = [c_vp]
    L.p2p_job_set_pano.restype = c_int
    L.p2p_job_set_pano.argtypes = [c_vp, c_int, c_vp, c_i64]
    L.p2p_job_set_pano_async.restype = c_int
    L.p2p_job_set_pano_async.argtypes = [c_vp, c_int, c_vp, c_i64]
    L.p2p_job_share_panos.restype = c_int
    L.p2p_job_share_panos.argtypes = [c_vp, c_vp]
    L.p2p_job_get_views_async.restype = c_int
    L.p2p_job_get_views_async.argtypes = [c_vp, c_int, c_vp]
    L.p2p_job_wait.restype = c_int
    L.p2p_job_wait.argtypes = [c_vp]
    L.p2p_job_set_maps.restype = c_int
    L.p2p_job_set_maps.argtypes = [c_vp, c_vp, c_vp, c_vp]
    L.p2p_job_run.restype = c_int
    L.p2p_job_run.argtypes = [c_vp]
    L.p2p_job_get_views.restype = c_int
    L.p2p_job_get_views.argtypes = [c_vp, c_int, c_vp]
    L.p2p_job_plan_ms.restype = c_int
    L.p2p_job_plan_ms.argtypes = [c_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.p2p_job_kernel_ms.restype = c_int
    L.p2p_job_kernel_ms.argtypes = [c_vp, ctypes.POINTER(ctypes.c_float)]
    L.p2p_job_kernel_ms_last.restype = c_int
    L.p2p_job_kernel_ms_last.argtypes = [c_vp, c_vp, c_int]
    L.p2p_job_device_out.restype = c_vp
    L.p2p_job_device_out.argtypes = [c_vp, ctypes.POINTER(c_i64)]
    L.p2p_job_get_coords.restype = c_int
    L.p2p_job_get_coords.argtypes = [c_vp, c_vp]
    L.p2p_job_get_yaw_tables.restype = c_int
    L.p2p_job_get_yaw_tables.argtypes = [c_vp, c_vp]
    L.p2p_job_set_yaws.restype = c_int
    L.p2p_job_set_yaws.argtypes = [c_vp, c_vp]
    L.p2p_host_alloc.restype = c_int
    L.p2p_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(c_vp)]
    L.p2p_host_free.restype = c_int
    L.p2p_host_free.argtypes = [c_vp]
    L.p2p_release_cache.restype = c_int
    L.p2p_release_cache.argtypes = []
    L.p2p_reload_options.restype = c_int
    L.p2p_reload_options.argtypes = []
    L.p2p_job_get_view.restype = c_int
    L.p2p_job_get_view.argtypes = [c_vp, c_int, c_int, c_int, c_vp]
    L.p2p_job_get_view_async.restype = c_int
    L.p2p_job_get_view_async.argtypes = [c_vp, c_int, c_int, c_int, c_vp]
    L.p2p_job_set_view_mask.restype = c_int
    L.p2p_job_set_view_mask.argtypes = [c_vp, c_vp]
    L.p2p_job_set_border.restype = c_int
    L.p2p_job_set_border.argtypes = [c_vp, c_int]
    L.p2p_job_set_rows.restype = c_int
    L.p2p_job_set_rows.argtypes = [c_vp, c_int, c_int]
    L.p2p_job_get_view_rows.restype = c_int
    L.p2p_job_get_view_rows.argtypes = [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]
    L.p2p_job_get_view_rows_async.restype = c_int
    L.p2p_job_get_view_rows_async.argtypes = [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]
    L.p2p_device_mem_info.restype = c_int
    L.p2p_device_mem_info.argtypes = [c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]
    L.p2p_job_get_info.restype = c_int
    L.p2p_job_get_info.argtypes = [c_vp, ctypes.POINTER(JobInfo)]
    _lib = L
    return L


def check(rc):
    if rc != P2P_OK:
        raise P2PError(rc, lib().p2p_last_error().decode("utf-8", "replace"))


def device_count():
    return lib().p2p_device_count()


def device_mem_info(device=0):
    """(free, total) bytes of device memory as the driver reports them."""
    f, t = ctypes.c_int64(), ctypes.c_int64()
    check(lib().p2p_device_mem_info(int(device), ctypes.byref(f), ctypes.byref(t)))
    return f.value, t.value


def reload_options():
    """Re-read the P2P_* environment variables (the library reads them once per process and copies them into a job when
    it is created).  Only while no other thread is inside the library: tests and tools that flip a knob between jobs."""
    check(lib().p2p_reload_options())


def version():
    return lib().p2p_version().decode()


def _i32(seq):
    a = np.ascontiguousarray(np.asarray(seq, dtype=np.int64))
    if a.ndim != 1:
        raise ValueError("angle lists must be one-dimensional")
    if a.size and (a.min() < -(2**31) or a.max() >= 2**31):
        raise ValueError("angle out of int32 range")
    return a.astype(np.int32)


def _f64(seq):
    """Angle lists for the real-valued entry points: any finite numbers of degrees (np.radians takes them, P:85, P:64-68)."""
    a = np.ascontiguousarray(np.asarray(seq, dtype=np.float64))
    if a.ndim != 1:
        raise ValueError("angle lists must be one-dimensional")
    if a.size and not np.isfinite(a).all():
        raise ValueError("angles must be finite")
    return a


def as_image(a, what="image"):
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise TypeError("%s must be uint8 (got %s)" % (what, a.dtype))
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("%s must have shape (H, W, 3), got %s" % (what, a.shape))
    if a.strides[2] != 1 or a.strides[1] != 3 or a.strides[0] < 3 * a.shape[1]:
        a = np.ascontiguousarray(a)
    return a


class JobDescF64(ctypes.Structure):
    _fields_ = [
        ("pw", ctypes.c_int32), ("ph", ctypes.c_int32), ("n_panos", ctypes.c_int32),
        ("n_yaw", ctypes.c_int32), ("yaw_deg", ctypes.POINTER(ctypes.c_double)),
        ("n_pitch", ctypes.c_int32), ("pitch_deg", ctypes.POINTER(ctypes.c_double)),
        ("fov_deg", ctypes.c_double), ("ow", ctypes.c_int32), ("oh", ctypes.c_int32),
        ("flags", ctypes.c_int32),
    ]


class _PinnedPool:
    """Free page-locked blocks kept for reuse (hipHostMalloc costs far more than the copy it speeds up):
    exact-size buckets, bounded by P2P_PINNED_POOL_MB (default 2048) of idle memory."""

    def __init__(self):
        import threading

        self.lock = threading.Lock()
        self.free = {}
        self.idle_bytes = 0
        self.live_bytes = 0   # handed out and not yet returned
        self.cap = int(os.environ.get("P2P_PINNED_POOL_MB", "2048")) << 20
        # page-locked memory is a limited resource: beyond this much in use, callers get ordinary arrays
        self.max_live = int(os.environ.get("P2P_PINNED_MAX_MB", "8192")) << 20

    def take(self, nbytes):
        with self.lock:
            if self.live_bytes + nbytes > self.max_live:
                raise MemoryError("page-locked memory budget (P2P_PINNED_MAX_MB) exhausted")
            self.live_bytes += nbytes
            lst = self.free.get(nbytes)
            if lst:
                self.idle_bytes -= nbytes
                return lst.pop()
        ptr = ctypes.c_void_p()
        try:
            check(lib().p2p_host_alloc(int(nbytes), ctypes.byref(ptr)))
        except Exception:
            with self.lock:
                self.live_bytes -= nbytes
            raise
        return ptr.value

    def give(self, ptr, nbytes):
        with self.lock:
            self.live_bytes -= nbytes
            if self.idle_bytes + nbytes <= self.cap:
                self.free.setdefault(nbytes, []).append(ptr)
                self.idle_bytes += nbytes
                return
        lib().p2p_host_free(ctypes.c_void_p(ptr))

    def trim(self):
        with self.lock:
            blocks = [p for lst in self.free.values() for p in lst]
            self.free.clear()
            self.idle_bytes = 0
        for p in blocks:
            lib().p2p_host_free(ctypes.c_void_p(p))


_pool = _PinnedPool()


class _PinnedBlock:
    """Owner of one page-locked block; it returns to the pool when the last array over it is collected."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = _pool.take(self.nbytes)

    def __del__(self):
        try:
            if self.ptr:
                _pool.give(self.ptr, self.nbytes)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.uint8):
    """np.empty in page-locked host memory (p2p_host_alloc): uploads from it and downloads into it run as
    DMA at PCIe rate.  The memory is released when the array and every view of it are gone."""
    dtype = np.dtype(dtype)
    shape = tuple(int(x) for x in (shape if np.iterable(shape) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    block = _PinnedBlock(max(nbytes, 1))
    buf = (ctypes.c_ubyte * max(nbytes, 1)).from_address(block.ptr)
    buf._p2p_owner = block  # arr.base -> buf -> block
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)


def release_cache():
    """Free the device buffers the one-shot calls of THIS thread keep between calls, and the idle
    page-locked blocks of the host pool."""
    check(lib().p2p_release_cache())
    _pool.trim()


def remap_views(pano, yaw_deg, pitch_deg, fov_deg, ow, oh, device=0, pinned=False, flags=0):
    """p2p_remap_views_u8 -> uint8 [n_yaw][n_pitch][oh][ow][3] (in page-locked memory if pinned)."""
    pano = as_image(pano, "pano_image")
    yaw, pitch = _i32(yaw_deg), _i32(pitch_deg)
    ph, pw = pano.shape[:2]
    shape = (yaw.size, pitch.size, int(oh), int(ow), 3)
    out = None
    if pinned and yaw.size and pitch.size:
        try:
            out = pinned_empty(shape)
        except (MemoryError, P2PError, OSError):  # no page-locked memory to be had: an ordinary array is merely slower to fill
            out = None
    if out is None:
        out = np.empty(shape, dtype=np.uint8)
    check(lib().p2p_remap_views_u8(pano.ctypes.data, pw, ph, pano.strides[0],
                                   yaw.ctypes.data, yaw.size, pitch.ctypes.data, pitch.size,
                                   int(fov_deg), int(ow), int(oh), out.ctypes.data, int(device), int(flags)))
    return out


def remap_views_f64(pano, yaw_deg, pitch_deg, fov_deg, ow, oh, device=0, pinned=False, flags=0):
    """p2p_remap_views_f64: as remap_views with real-valued yaw / pitch / FOV (what the reference's functions accept)."""
    pano = as_image(pano, "pano_image")
    yaw, pitch = _f64(yaw_deg), _f64(pitch_deg)
    ph, pw = pano.shape[:2]
    shape = (yaw.size, pitch.size, int(oh), int(ow), 3)
    out = None
    if pinned and yaw.size and pitch.size:
        try:
            out = pinned_empty(shape)
        except (MemoryError, P2PError, OSError):  # no page-locked memory to be had: an ordinary array is merely slower to fill
            out = None
    if out is None:
        out = np.empty(shape, dtype=np.uint8)
    check(lib().p2p_remap_views_f64(pano.ctypes.data, pw, ph, pano.strides[0],
                                    yaw.ctypes.data, yaw.size, pitch.ctypes.data, pitch.size,
                                    float(fov_deg), int(ow), int(oh), out.ctypes.data, int(device), int(flags)))
    return out


def remap_views_maps(pano, yaw_rows, U, V, device=0):
    """p2p_remap_views_maps_u8: caller float maps.  yaw_rows [n_yaw][pw]; U, V [n_pitch][oh][ow]."""
    pano = as_image(pano, "pano_image")
    ph, pw = pano.shape[:2]
    yaw_rows = np.ascontiguousarray(yaw_rows, dtype=np.float32)
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    if yaw_rows.ndim != 2 or yaw_rows.shape[1] != pw:
        raise ValueError("yaw_rows must be [n_yaw][pano_width]")
    if U.ndim != 3 or U.shape != V.shape:
        raise ValueError("U and V must both be [n_pitch][oh][ow]")
    n_pitch, oh, ow = U.shape
    out = np.empty((yaw_rows.shape[0], n_pitch, oh, ow, 3), dtype=np.uint8)
    check(lib().p2p_remap_views_maps_u8(pano.ctypes.data, pw, ph, pano.strides[0],
                                        yaw_rows.ctypes.data, yaw_rows.shape[0],
                                        U.ctypes.data, V.ctypes.data, n_pitch, ow, oh,
                                        out.ctypes.data, int(device)))
    return out


def remap_views_pitch_maps(pano, yaw_deg, U, V, maps_key=0, device=0, pinned=False):
    """p2p_remap_views_pitch_maps_f64: yaw tables from yaw_deg on the device, caller PITCH maps U, V [n_pitch][oh][ow]
    (the tool's --exact route).  maps_key names exactly these maps (0: none): a slot that holds them is not sent them again."""
    pano = as_image(pano, "pano_image")
    ph, pw = pano.shape[:2]
    yaw = _f64(yaw_deg)
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    if U.ndim != 3 or U.shape != V.shape:
        raise ValueError("U and V must both be [n_pitch][oh][ow]")
    n_pitch, oh, ow = U.shape
    shape = (yaw.size, n_pitch, oh, ow, 3)
    out = None
    if pinned and yaw.size and n_pitch:
        try:
            out = pinned_empty(shape)
        except (MemoryError, P2PError, OSError):
            out = None
    if out is None:
        out = np.empty(shape, dtype=np.uint8)
    check(lib().p2p_remap_views_pitch_maps_f64(pano.ctypes.data, pw, ph, pano.strides[0], yaw.ctypes.data, yaw.size,
                                               U.ctypes.data, V.ctypes.data, n_pitch, int(maps_key), ow, oh,
                                               out.ctypes.data, int(device)))
    return out


def remap_maps(src, U, V, border=BORDER_CONSTANT, border_value=None, device=0, interpolation=INTER_LINEAR):
    """p2p_remap_maps_interp_u8 == cv2.remap(src, U, V, interpolation, borderMode=border)."""
    src = np.asarray(src)
    if src.dtype != np.uint8:
        raise TypeError("src must be uint8")
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    if src.ndim != 3 or src.shape[2] not in (1, 3, 4):
        raise ValueError("src must be (H, W), (H, W, 1), (H, W, 3) or (H, W, 4)")
    src = np.ascontiguousarray(src)
    sh, sw, cn = src.shape
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    if U.ndim != 2 or U.shape != V.shape:
        raise ValueError("U and V must be two 2-D arrays of one shape")
    oh, ow = U.shape
    out = np.empty((oh, ow, cn), dtype=np.uint8)
    bv = None
    if border_value is not None:
        bv = np.zeros(4, dtype=np.uint8)
        bv[:cn] = np.asarray(border_value, dtype=np.uint8).ravel()[:cn]
    check(lib().p2p_remap_maps_interp_u8(src.ctypes.data, sw, sh, src.strides[0], cn, U.ctypes.data, V.ctypes.data,
                                         ow, oh, out.ctypes.data, int(interpolation), int(border),
                                         None if bv is None else bv.ctypes.data, int(device)))
    return out[:, :, 0] if squeeze else out


def remap_maps_batch(src, U, V, border=BORDER_CONSTANT, device=0):
    """p2p_remap_maps_batch_u8: cv2.remap(src, U[k], V[k], INTER_LINEAR, borderMode=border) for every k, one launch."""
    src = as_image(src, "src")
    sh, sw = src.shape[:2]
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    if U.ndim != 3 or U.shape != V.shape:
        raise ValueError("U and V must both be [n_maps][oh][ow]")
    n, oh, ow = U.shape
    out = np.empty((n, oh, ow, 3), dtype=np.uint8)
    check(lib().p2p_remap_maps_batch_u8(src.ctypes.data, sw, sh, src.strides[0], U.ctypes.data, V.ctypes.data,
                                        n, ow, oh, out.ctypes.data, int(border), int(device)))
    return out


def build_pitch_map(ow, oh, fov_rad, pitch_rad, pw, ph, device=0):
    U = np.empty((int(oh), int(ow)), dtype=np.float32)
    V = np.empty_like(U)
    check(lib().p2p_build_pitch_map(int(ow), int(oh), float(fov_rad), float(pitch_rad), int(pw), int(ph),
                                    U.ctypes.data, V.ctypes.data, int(device)))
    return U, V


def build_rot_map(ow, oh, fov_rad, R, pw, ph, device=0):
    """p2p_build_rot_map: the legacy tool's combined-rotation map for a float32 3x3 matrix R."""
    R = np.ascontiguousarray(R, dtype=np.float32)
    if R.shape != (3, 3):
        raise ValueError("R must be 3x3")
    U = np.empty((int(oh), int(ow)), dtype=np.float32)
    V = np.empty_like(U)
    check(lib().p2p_build_rot_map(int(ow), int(oh), float(fov_rad), R.ctypes.data, int(pw), int(ph),
                                  U.ctypes.data, V.ctypes.data, int(device)))
    return U, V


def build_yaw_row(pw, yaw_rad, device=0):
    row = np.empty(int(pw), dtype=np.float32)
    check(lib().p2p_build_yaw_row(int(pw), float(yaw_rad), row.ctypes.data, int(device)))
    return row


def _finalizing():
    import sys

    return sys is None or sys.is_finalizing()


class Context:
    """p2p_ctx: one device, one HIP stream."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        check(lib().p2p_ctx_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)

    def synchronize(self):
        check(lib().p2p_ctx_synchronize(self._h))

    def mark(self, which):
        check(lib().p2p_ctx_mark(self._h, int(which)))

    def marked_ms(self):
        ms = ctypes.c_float()
        check(lib().p2p_ctx_marked_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def close(self):
        if self._h:
            lib().p2p_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            if not _finalizing():  # no HIP call once the interpreter (and with it the runtime) is going down
                self.close()
        except Exception:
            pass


class Job:
    """p2p_job: n_panos resident panoramas of one size x (yaw x pitch) views, outputs resident in HBM."""

    def __init__(self, ctx, pw, ph, n_panos, yaw_deg, pitch_deg, fov_deg, ow, oh, flags=0, integer_abi=False):
        self.ctx = ctx
        self._h = ctypes.c_void_p()
        if integer_abi:  # p2p_job_create: integer degrees, pitch checked against 1..179 as the CLI does
            self._yaw, self._pitch = _i32(yaw_deg), _i32(pitch_deg)
            d = JobDesc(int(pw), int(ph), int(n_panos),
                        self._yaw.size, self._yaw.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                        self._pitch.size, self._pitch.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                        int(fov_deg), int(ow), int(oh), int(flags))
            check(lib().p2p_job_create(ctx._h, ctypes.byref(d), ctypes.byref(self._h)))
        else:
            self._yaw, self._pitch = _f64(yaw_deg), _f64(pitch_deg)
            d = JobDescF64(int(pw), int(ph), int(n_panos),
                           self._yaw.size, self._yaw.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                           self._pitch.size, self._pitch.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                           float(fov_deg), int(ow), int(oh), int(flags))
            check(lib().p2p_job_create_f64(ctx._h, ctypes.byref(d), ctypes.byref(self._h)))
        self.pw, self.ph, self.n_panos = int(pw), int(ph), int(n_panos)
        self.n_yaw, self.n_pitch, self.ow, self.oh = self._yaw.size, self._pitch.size, int(ow), int(oh)
        self._inflight = []  # host arrays of asynchronous copies still in flight
        self._owner = None

    def set_pano(self, index, pano, wait=True):
        """wait=False: the upload is only enqueued (p2p_job_set_pano_async); the array is kept alive here until
        wait() / the next synchronous call, and must not be modified meanwhile."""
        pano = as_image(pano, "pano_image")
        if pano.shape[:2] != (self.ph, self.pw):
            raise ValueError("panorama is %s, job expects (%d, %d)" % (pano.shape[:2], self.ph, self.pw))
        if wait:
            check(lib().p2p_job_set_pano(self._h, int(index), pano.ctypes.data, pano.strides[0]))
        else:
            check(lib().p2p_job_set_pano_async(self._h, int(index), pano.ctypes.data, pano.strides[0]))
            self._inflight.append(pano)

    def share_panos(self, owner):
        check(lib().p2p_job_share_panos(self._h, owner._h))
        self._owner = owner  # keeps it alive

    def wait(self):
        check(lib().p2p_job_wait(self._h))
        self._inflight.clear()

    def set_yaws(self, yaw_deg):
        yaw = _f64(yaw_deg)
        if yaw.size != self.n_yaw:
            raise ValueError("the job was created with %d yaws, got %d" % (self.n_yaw, yaw.size))
        check(lib().p2p_job_set_yaws_f64(self._h, yaw.ctypes.data))
        self._yaw = yaw

    def set_maps(self, yaw_rows, U, V):
        U = np.ascontiguousarray(U, dtype=np.float32)
        V = np.ascontiguousarray(V, dtype=np.float32)
        if U.shape != (self.n_pitch, self.oh, self.ow) or V.shape != U.shape:
            raise ValueError("U, V must be [n_pitch][oh][ow]")
        rows_p = None
        if yaw_rows is not None:
            yaw_rows = np.ascontiguousarray(yaw_rows, dtype=np.float32)
            if yaw_rows.shape != (self.n_yaw, self.pw):
                raise ValueError("yaw_rows must be [n_yaw][pw]")
            rows_p = yaw_rows.ctypes.data
        check(lib().p2p_job_set_maps(self._h, rows_p, U.ctypes.data, V.ctypes.data))

    def run(self):
        check(lib().p2p_job_run(self._h))

    def time_launches(self, n):
        """Launch timing is off by default.  n launches to keep event pairs for (True = 256), 0 / False = off."""
        check(lib().p2p_job_time_launches(self._h, 256 if n is True else int(n)))

    def info(self):
        """p2p_job_get_info as a dict: tile shape, pairs per workgroup, list order, gather tiles ..."""
        i = JobInfo()
        check(lib().p2p_job_get_info(self._h, ctypes.byref(i)))
        return {k: getattr(i, k) for k, _ in JobInfo._fields_ if k != "reserved"}

    def plan_ms(self):
        """(plan pass, yaw tables) build times in ms of the tables this job uses (built once per geometry and context)."""
        a, b = ctypes.c_float(), ctypes.c_float()
        check(lib().p2p_job_plan_ms(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def kernel_ms(self):
        ms = ctypes.c_float()
        check(lib().p2p_job_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def kernel_ms_last(self, n):
        out = np.empty(int(n), dtype=np.float32)
        check(lib().p2p_job_kernel_ms_last(self._h, out.ctypes.data, int(n)))
        return out

    def get_views(self, index=0, pinned=False):
        shape = (self.n_yaw, self.n_pitch, self.oh, self.ow, 3)
        out = None
        if pinned:
            try:
                out = pinned_empty(shape)
            except (MemoryError, P2PError, OSError):
                out = None
        if out is None:
            out = np.empty(shape, dtype=np.uint8)
        check(lib().p2p_job_get_views(self._h, int(index), out.ctypes.data))
        return out

    def set_view_mask(self, mask):
        """mask: bool / uint8 [n_yaw][n_pitch] of the views the job draws, or None for all (p2p_job_set_view_mask)."""
        if mask is None:
            check(lib().p2p_job_set_view_mask(self._h, None))
            return
        m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        if m.shape != (self.n_yaw, self.n_pitch):
            raise ValueError("view mask must be [n_yaw][n_pitch] = (%d, %d), got %s" % (self.n_yaw, self.n_pitch, m.shape))
        check(lib().p2p_job_set_view_mask(self._h, m.ctypes.data))

    def set_border(self, border):
        """cv2 border code of the job's pitch stage (p2p_job_set_border): BORDER_REFLECT makes a resident legacy-tool job."""
        check(lib().p2p_job_set_border(self._h, int(border)))

    def set_rows(self, row0, row1):
        """The job draws only output rows [row0, row1) of every view (whole tile rows of 16; p2p_job_set_rows)."""
        check(lib().p2p_job_set_rows(self._h, int(row0), int(row1)))

    def get_view_rows(self, yaw_i, pitch_i, row0, row1, index=0):
        """Rows [row0, row1) of one view, [rows][ow][3] (p2p_job_get_view_rows)."""
        out = np.empty((int(row1) - int(row0), self.ow, 3), dtype=np.uint8)
        check(lib().p2p_job_get_view_rows(self._h, int(index), int(yaw_i), int(pitch_i), int(row0), int(row1), out.ctypes.data))
        return out

    def get_view_rows_async(self, yaw_i, pitch_i, row0, row1, out, index=0):
        """Enqueue the download of rows [row0, row1) of one view into `out` ([rows][ow][3] uint8, C-contiguous); complete after wait()."""
        if out.dtype != np.uint8 or out.shape != (int(row1) - int(row0), self.ow, 3) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous uint8 [row1 - row0][ow][3] array")
        check(lib().p2p_job_get_view_rows_async(self._h, int(index), int(yaw_i), int(pitch_i), int(row0), int(row1), out.ctypes.data))
        self._inflight.append(out)

    def get_view_async(self, yaw_i, pitch_i, out, index=0):
        """Enqueue the download of one view into `out` ([oh][ow][3] uint8, C-contiguous); complete after wait()."""
        if out.dtype != np.uint8 or out.shape != (self.oh, self.ow, 3) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous uint8 [oh][ow][3] array")
        check(lib().p2p_job_get_view_async(self._h, int(index), int(yaw_i), int(pitch_i), out.ctypes.data))
        self._inflight.append(out)

    def get_view(self, yaw_i, pitch_i, index=0):
        """One view [oh][ow][3] of panorama `index` (p2p_job_get_view)."""
        out = np.empty((self.oh, self.ow, 3), dtype=np.uint8)
        check(lib().p2p_job_get_view(self._h, int(index), int(yaw_i), int(pitch_i), out.ctypes.data))
        return out

    def get_views_async(self, index=0, out=None, pinned=True):
        """Enqueue the download (p2p_job_get_views_async); the returned array is complete after wait()."""
        shape = (self.n_yaw, self.n_pitch, self.oh, self.ow, 3)
        if out is None:
            try:
                out = pinned_empty(shape) if pinned else np.empty(shape, dtype=np.uint8)
            except (MemoryError, P2PError, OSError):
                out = np.empty(shape, dtype=np.uint8)
        check(lib().p2p_job_get_views_async(self._h, int(index), out.ctypes.data))
        self._inflight.append(out)
        return out

    def get_coords(self):
        out = np.empty((self.n_pitch, self.oh, self.ow, 2), dtype=np.int32)
        check(lib().p2p_job_get_coords(self._h, out.ctypes.data))
        return out

    def get_yaw_tables(self):
        out = np.empty((self.n_yaw, self.pw), dtype=np.uint32)
        check(lib().p2p_job_get_yaw_tables(self._h, out.ctypes.data))
        return out

    def device_out(self):
        n = ctypes.c_int64()
        p = lib().p2p_job_device_out(self._h, ctypes.byref(n))
        return p, n.value

    def close(self):
        if self._h:
            lib().p2p_job_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            if not _finalizing():
                self.close()
        except Exception:
            pass
