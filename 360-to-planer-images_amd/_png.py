"""PNG files the way the reference's writer makes them.

The reference writes every view with cv2.imwrite(path, image) and no parameters
(/root/reference/app/panorama_to_plane-pitch.py:277, /root/reference/app/legacy/panorama_to_plane.py:275).  For .png
OpenCV 4.10 (opencv-python 4.10.0.84, modules/imgcodecs/src/grfmt_png.cpp: PngEncoder::write; a third-party dependency,
not in /root/reference) then tunes libpng for speed: filter type SUB on every row, zlib level Z_BEST_SPEED, strategy Z_RLE,
8 bits per sample, no interlace.  This module writes that stream with zlib and NumPy: the same filter, level and strategy
-- the pixels a reader gets back are the array's, as with any PNG; the bytes of the compressed stream are zlib's, not
libpng's buffer-by-buffer chunking of them (one IDAT here).

Why not Pillow: its encoder picks a filter per row by trial (the PNG specification's heuristic) and deflates with the
default strategy -- 56 ms for an 800 x 800 view at compress_level=1 where this takes 24 (noise: 113 against 21), and the
encoder is what the tool waits for once the resampling is on the GPU (profiles/r06_cli_end_to_end.txt).

Host-side file format code: no pixel of a view is computed here.
"""
import struct
import zlib

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"
_COLOUR_TYPE = {1: 0, 2: 4, 3: 2, 4: 6}  # channels -> PNG colour type (grey, grey + alpha, RGB, RGBA)


def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(data, zlib.crc32(kind)) & 0xFFFFFFFF)


def encode_png(image):
    """bytes of a PNG file for a uint8 array (H, W), (H, W, 1), (H, W, 2), (H, W, 3: RGB) or (H, W, 4: RGBA).
    Raises ValueError for anything else (callers fall back to their general encoder)."""
    a = np.asarray(image)
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or a.size == 0:
        raise ValueError("encode_png: a non-empty uint8 array of 2 or 3 dimensions is required")
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, cn = a.shape
    if cn not in _COLOUR_TYPE or h >= 1 << 31 or w >= 1 << 31:
        raise ValueError("encode_png: 1 to 4 channels")
    flat = np.ascontiguousarray(a).reshape(h, w * cn)
    # every row: filter type byte 1 (SUB), then each byte minus the byte one pixel to its left (modulo 256; the first
    # pixel of a row has no left neighbour and is stored as it is)
    raw = np.empty((h, 1 + w * cn), dtype=np.uint8)
    raw[:, 0] = 1
    raw[:, 1:1 + cn] = flat[:, :cn]
    if w > 1:
        np.subtract(flat[:, cn:], flat[:, :-cn], out=raw[:, 1 + cn:])
    z = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_RLE)  # Z_BEST_SPEED, libpng's window and memory level, Z_RLE
    data = z.compress(raw) + z.flush()
    ihdr = struct.pack(">IIBBBBB", w, h, 8, _COLOUR_TYPE[cn], 0, 0, 0)
    return b"".join((_SIGNATURE, _chunk(b"IHDR", ihdr), _chunk(b"IDAT", data), _chunk(b"IEND", b"")))
