"""oracle/cpu_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of oracle/cv_remap_oracle.c (the restated cv2.remap fixed-point
gather; PARITY UNPINNED, see that file's header) plus the reference's two-stage
view synthesis restated on top of it:

  process_yaw_and_pitchs : /root/reference/app/panorama_to_plane-pitch.py:181-221
  panorama_to_plane      : /root/reference/app/legacy/panorama_to_plane.py:159-194

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product never does.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import maps

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "cv_remap_oracle.c")
_LIB = os.path.join(_HERE, "libp2p_oracle.so")

BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4

_lib = None


def build(force=False):
    """gcc the C restatement into oracle/libp2p_oracle.so (no fast-math, no FMA contraction)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        tmp = _LIB + ".tmp.%d" % os.getpid()
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
             "-Wall", "-o", tmp, _SRC, "-lm"]
        )
        os.replace(tmp, _LIB)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.orc_remap_u8.restype = ctypes.c_int
        L.orc_remap_u8.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
            ctypes.c_int, ctypes.c_void_p,
        ]
        L.orc_quantise_maps.restype = None
        L.orc_quantise_maps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_void_p]
        L.orc_get_wtab.restype = None
        L.orc_get_wtab.argtypes = [ctypes.c_void_p]
        for fn in (L.orc_remap_nearest_u8, L.orc_remap_cubic_u8):
            fn.restype = ctypes.c_int
            fn.argtypes = L.orc_remap_u8.argtypes
        L.orc_get_cubic_wtab.restype = None
        L.orc_get_cubic_wtab.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def weight_table():
    out = np.zeros((1024, 4), dtype=np.int16)
    lib().orc_get_wtab(out.ctypes.data)
    return out


def quantise_maps(U, V):
    """(ix, iy) int16 and (fx, fy) 0..31 exactly as cv::remap's RemapInvoker derives them."""
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    ixy = np.empty(U.shape + (2,), dtype=np.int16)
    fxy = np.empty(U.shape, dtype=np.uint16)
    lib().orc_quantise_maps(U.ctypes.data, V.ctypes.data, U.size, ixy.ctypes.data, fxy.ctypes.data)
    return ixy[..., 0], ixy[..., 1], (fxy & 31).astype(np.int32), (fxy >> 5).astype(np.int32)


INTER_NEAREST, INTER_LINEAR, INTER_CUBIC = 0, 1, 2  # cv2's codes


def cubic_weight_table():
    out = np.zeros((1024, 16), dtype=np.int16)
    lib().orc_get_cubic_wtab(out.ctypes.data)
    return out


def remap(src, U, V, border=BORDER_CONSTANT, border_value=None, interpolation=INTER_LINEAR):
    """cv2.remap(src, U, V, interpolation, borderMode=border, borderValue=border_value) for uint8."""
    src = np.asarray(src)
    if src.dtype != np.uint8:
        raise TypeError("oracle remap handles uint8 only")
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    if not src.flags.c_contiguous:
        src = np.ascontiguousarray(src)
    sh, sw, cn = src.shape
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    if U.shape != V.shape or U.ndim != 2:
        raise ValueError("maps must be two 2-D float32 arrays of one shape")
    dh, dw = U.shape
    dst = np.empty((dh, dw, cn), dtype=np.uint8)
    bv = None
    if border_value is not None:
        bv = np.zeros(4, dtype=np.uint8)
        bv[:cn] = np.asarray(border_value, dtype=np.uint8).ravel()[:cn]
    fn = {INTER_NEAREST: lib().orc_remap_nearest_u8, INTER_LINEAR: lib().orc_remap_u8,
          INTER_CUBIC: lib().orc_remap_cubic_u8}[interpolation]
    rc = fn(
        src.ctypes.data, sw, sh, src.strides[0], cn,
        U.ctypes.data, V.ctypes.data, dw,
        dst.ctypes.data, dw, dh, dst.strides[0],
        border, None if bv is None else bv.ctypes.data,
    )
    if rc != 0:
        raise ValueError("oracle remap rejected its arguments (rc=%d)" % rc)
    return dst[:, :, 0] if squeeze else dst


def yaw_stage(pano_image, yaw_angle):
    """Stage 1 of P:181-221 (P:191-199): the whole panorama resampled by the yaw map."""
    ph, pw = pano_image.shape[:2]
    U, V = maps.yaw_map(pw, ph, yaw_angle)
    return remap(pano_image, U, V, BORDER_CONSTANT)


def process_yaw_and_pitchs(pano_image, yaw_angle, pitch_angles, output_width, output_height, fov_deg=90,
                           _pitch_cache=None):
    """P:181-221 restated: yaw remap of the panorama, then one pitch remap per pitch angle."""
    ph, pw = pano_image.shape[:2]
    rotated = yaw_stage(pano_image, yaw_angle)
    slices = []
    for pitch in pitch_angles:
        key = (output_width, output_height, pitch, pw, ph, fov_deg)
        if _pitch_cache is not None and key in _pitch_cache:
            U, V = _pitch_cache[key]
        else:
            U, V = maps.pitch_map_deg(output_width, output_height, pitch, pw, ph, fov_deg)
            if _pitch_cache is not None:
                _pitch_cache[key] = (U, V)
        slices.append(remap(rotated, U, V, BORDER_CONSTANT))
    return slices


def panorama_to_plane(pano_array, U, V):
    """L:182-194 -> interpolate_color(U, V, img, 'bilinear') -> cv2.remap(..., BORDER_REFLECT) (L:179)."""
    return remap(pano_array, U, V, BORDER_REFLECT)
