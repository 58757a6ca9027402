"""The package's PNG writer (360-to-planer-images_amd/_png.py: cv2.imwrite's default settings -- SUB filter, Z_BEST_SPEED,
Z_RLE -- written with zlib and NumPy) against an independent reader: whatever array goes in comes back from Pillow's
decoder, for every channel count, odd sizes, one-pixel rows and columns, strided and channel-reversed views; the header says
what the array is; anything the writer does not take raises ValueError (the tools then use Pillow's encoder)."""
import importlib
import io
import struct
import zlib

import numpy as np
import pytest

png = importlib.import_module("360-to-planer-images_amd._png")


def _decode(data):
    from PIL import Image
    with Image.open(io.BytesIO(data)) as im:
        im.load()
        return np.asarray(im), im.mode


@pytest.mark.parametrize("cn,mode", [(0, "L"), (1, "L"), (2, "LA"), (3, "RGB"), (4, "RGBA")])
def test_round_trip_through_pillow(cn, mode):
    rng = np.random.default_rng(17 + cn)
    for h, w in ((1, 1), (1, 7), (9, 1), (33, 70), (128, 257), (800, 800)):
        shape = (h, w) if cn == 0 else (h, w, cn)
        for kind in ("noise", "smooth"):
            if kind == "noise":
                a = rng.integers(0, 256, shape, dtype=np.uint8)
            else:
                a = (np.add.outer(np.arange(h) * 3, np.arange(w) * 5) % 256).astype(np.uint8)
                a = a if cn == 0 else np.stack([(a + 40 * c).astype(np.uint8) for c in range(cn)], axis=-1)
            back, got_mode = _decode(png.encode_png(a))
            assert got_mode == mode
            assert np.array_equal(back.reshape(a.shape), a), (shape, kind)


def test_views_of_other_arrays_and_the_header():
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    for view in (big[:, :, ::-1], big[3:31:2, 5:50:3], np.asfortranarray(big)):
        back, _ = _decode(png.encode_png(view))
        assert np.array_equal(back, view)
    data = png.encode_png(big)
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and data[12:16] == b"IHDR"
    w, h, depth, colour, comp, filt, interlace = struct.unpack(">IIBBBBB", data[16:29])
    assert (w, h, depth, colour, comp, filt, interlace) == (60, 40, 8, 2, 0, 0, 0)
    # every chunk's CRC is right and the stream ends with IEND
    at, kinds = 8, []
    while at < len(data):
        n, kind = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        assert struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + body) & 0xFFFFFFFF
        kinds.append(kind)
        at += 12 + n
    assert kinds == [b"IHDR", b"IDAT", b"IEND"]
    # every row carries filter type 1 (SUB), as cv2.imwrite's default stream does
    rows = np.frombuffer(zlib.decompress(data[data.index(b"IDAT") + 4:-16]), np.uint8).reshape(40, 1 + 180)
    assert (rows[:, 0] == 1).all()


def test_what_the_writer_refuses():
    for bad in (np.zeros((4, 4, 3), np.uint16), np.zeros((4, 4, 5), np.uint8), np.zeros((0, 4, 3), np.uint8),
                np.zeros((4,), np.uint8), np.zeros((2, 2, 2, 2), np.uint8), np.zeros((4, 4), np.float32)):
        with pytest.raises(ValueError):
            png.encode_png(bad)
