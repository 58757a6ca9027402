#!/usr/bin/env python3
"""Host-side mirror of the reference tool app/panorama_to_plane-pitch.py ("P") on top of the
MI355X library libp2p_hip.so: same function names, argument order, defaults, file naming,
logging format and error behaviour, so a user of the reference can switch files.

    reference interface (P:line)                      here
    ------------------------------------------------  -------------------------------------------
    get_version                         P:22-27       get_version
    get_yaw_mapping / get_pitch_mapping P:42-73       same names, same cache keys (device-built maps)
    precompute_yaw_mapping              P:79-108      p2p_build_yaw_row  (one row, broadcast)
    precompute_pitch_mapping            P:114-175     p2p_build_pitch_map
    process_yaw_and_pitchs              P:181-221     p2p_remap_views_u8 (both remaps, one kernel)
    process_single_image                P:227-280     one resident job per image (all yaws x pitches)
    main                                P:286-356     same walk / logging / error swallowing
    check_pitch                         P:362-376     same messages
    CLI                                 P:382-488     same flags and defaults (+ additive --device, --devices,
                                                      --exact, --quality, --pixel_centres)

All pixel arithmetic runs on the GPU, and so do the maps by default.  There is no CPU fallback: without the built
library or without a HIP device the calls raise.

--exact / process_yaw_and_pitchs(..., exact=True) / set_exact(True): the identical-results mode.  The pitch maps are
evaluated on the host exactly as the reference evaluates them (P:114-175: NumPy float32 flow, libm, one sgemm; see
_exact_maps.py), cached under the reference's key (P:62), handed to the device once per geometry, and every pixel is
drawn from them by the same fixed-point HIP kernels: the reference's bytes on any panorama.  The default evaluates the
pitch maps on the device (within 1e-5 of the host's), which moves 0.001-0.017 % of cv2.remap's 1/32-px quantisations:
+-1 level on band-limited panoramas, more on hard edges.

Image files are decoded / encoded with Pillow (cv2 is not a dependency here).  At the API boundary arrays keep
cv2's BGR channel order because the reference hands BGR arrays around (P:243-244); the kernels are channel-agnostic,
so the file-to-file path (main / process_single_image) keeps Pillow's RGB order end to end and swaps nothing.
"""
import argparse
import logging
import os
import sys
from pathlib import Path

import numpy as np

if __package__ in (None, ""):  # executed as a script: make the sibling module importable
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _native  # type: ignore
else:
    from . import _native
try:
    from . import _driver, _exact_maps, _png
except ImportError:  # executed as a script
    import _driver  # type: ignore
    import _exact_maps  # type: ignore
    import _png  # type: ignore

VERSION = "0.3.2"  # the reference release this file mirrors (P:20)

# same module-level caches, same keys as P:17-18 / P:47 / P:62
yaw_mapping_cache = {}
pitch_mapping_cache = {}

_DEVICE = int(os.environ.get("P2P_DEVICE", "0"))
_DEVICES = None  # devices the directory walk of main() deals images to (None: just _DEVICE)
_EXACT = False   # set_exact / --exact: host-evaluated pitch maps, the reference's bytes
_QUALITY = "u8"  # set_quality / --quality: "u8" (the reference's arithmetic) | "f32" | "f16" (opt-in float resample)
_CENTRES = False  # set_quality(..., pixel_centres=True) / --pixel_centres: float paths only, sample through pixel centres


def set_device(device):
    """Select the HIP device used by the functions of this module (additive to the reference API)."""
    global _DEVICE
    _DEVICE = int(device)


def set_devices(devices):
    """Devices main() spreads a folder of images over, round-robin, one host thread and one HIP stream
    per device, no collective (every image / yaw / pitch view is independent work).
    None or an empty list restores single-device operation."""
    global _DEVICES
    _DEVICES = [int(d) for d in devices] if devices else None


def set_exact(on=True):
    """The identical-results mode for every later call of this module (the CLI's --exact): pitch maps as the reference
    computes them, on the host; pixels on the GPU from those maps.  Additive to the reference API."""
    global _EXACT
    if on and _QUALITY != "u8":
        raise ValueError("--exact is the reference's fixed-point arithmetic; the float pixel paths have no reference counterpart")
    _EXACT = bool(on)


def set_quality(pixel_path="u8", pixel_centres=False):
    """Pixel arithmetic of every later call (the CLI's --quality): "u8" = the reference's two fixed-point cv2.remap stages;
    "f32" / "f16" = ONE float resample per view with true wrap-around at the seam (not in the reference).
    pixel_centres (float paths only, --pixel_centres): rays through the centres of the output pixels and panorama texels
    centred at i + 0.5 -- the reference samples at integer coordinates (P:122-131), which shifts the picture by half a pixel."""
    global _QUALITY, _CENTRES
    if pixel_path not in _PIXEL_PATHS:
        raise ValueError(f"quality must be one of {sorted(_PIXEL_PATHS)}, got {pixel_path!r}")
    if pixel_path != "u8" and _EXACT:
        raise ValueError("--exact is the reference's fixed-point arithmetic; the float pixel paths have no reference counterpart")
    if pixel_centres and pixel_path == "u8":
        raise ValueError("pixel centres are a convention of the float pixel paths (--quality f32 / f16); u8 is the reference's arithmetic")
    _QUALITY, _CENTRES = pixel_path, bool(pixel_centres)


def get_version():
    return VERSION


# ----------------------------------------------------------------------------------------------
# coordinate maps (returned as float32 arrays exactly like the reference's)
# ----------------------------------------------------------------------------------------------
def precompute_yaw_mapping(pano_width, pano_height, yaw_angle):
    """(U, V) float32 (pano_height, pano_width) of P:79-108.  The formula depends on the column
    only, so the device computes one row (bit-exact dtype flow) and it is broadcast here."""
    logging.debug(f"[Yaw] Precomputing yaw mapping for yaw_angle: {yaw_angle} degrees")
    row = _native.build_yaw_row(pano_width, float(np.radians(yaw_angle)), _DEVICE)
    U = np.broadcast_to(row, (pano_height, pano_width)).copy()
    V = np.broadcast_to(np.arange(pano_height, dtype=np.float32)[:, None], (pano_height, pano_width)).copy()
    return U, V


def precompute_pitch_mapping(W, H, FOV_rad, pitch_radian, pano_width, pano_height):
    """(U, V) float32 (H, W) of P:114-175, evaluated by the same device function the fused kernel uses."""
    return _native.build_pitch_map(W, H, float(FOV_rad), float(pitch_radian), pano_width, pano_height, _DEVICE)


def get_yaw_mapping(pano_width, pano_height, yaw_angle):
    key = (pano_width, pano_height, yaw_angle)
    if key not in yaw_mapping_cache:
        yaw_mapping_cache[key] = precompute_yaw_mapping(pano_width, pano_height, yaw_angle)
    return yaw_mapping_cache[key]


def get_pitch_mapping(output_width, output_height, pitch_angle, pano_width, pano_height, fov_deg=90):
    if _EXACT:  # the maps the exact mode draws from: the reference's values on this host (their own cache, same key)
        return _exact_maps.get_pitch_mapping(output_width, output_height, pitch_angle, pano_width, pano_height, fov_deg)
    key = (output_width, output_height, pitch_angle, pano_width, pano_height, fov_deg)
    if key not in pitch_mapping_cache:
        pitch_mapping_cache[key] = precompute_pitch_mapping(
            W=output_width,
            H=output_height,
            FOV_rad=np.radians(fov_deg),
            pitch_radian=np.radians(pitch_angle),
            pano_width=pano_width,
            pano_height=pano_height,
        )
    return pitch_mapping_cache[key]


# ----------------------------------------------------------------------------------------------
# view synthesis
# ----------------------------------------------------------------------------------------------
def _angle(value, what):
    """Degrees as the reference's functions take them: any real number -- they go through np.radians (P:85,
    P:64-68); only the CLI narrows yaw / FOV to int and the pitch to 1..179 (check_pitch)."""
    if isinstance(value, (bool, np.bool_)) or not isinstance(value, (int, float, np.integer, np.floating)):
        raise TypeError(f"{what} must be a number of degrees, got {value!r}")
    v = float(value)
    if not np.isfinite(v):
        raise ValueError(f"{what} must be finite, got {value!r}")
    return v


_PINNED = os.environ.get("P2P_PINNED", "1") != "0"


_PIXEL_PATHS = {"u8": 0, "f32": _native.FLAG_PIXELS_F32, "f16": _native.FLAG_PIXELS_F16}


def _flags_of(pixel_path):
    return _PIXEL_PATHS[pixel_path] | (_native.FLAG_PIXEL_CENTRES if (_CENTRES and pixel_path != "u8") else 0)


def _exact_maps_for(pano_image, pitch_angles, output_width, output_height, fov_deg):
    """(U, V, maps_key) of the exact mode for this panorama's size: the reference's pitch maps (P:55-73, P:114-175),
    evaluated on the host once per key and named for the device."""
    shape = np.shape(pano_image)
    if len(shape) != 3:
        raise ValueError("pano_image must be (height, width, 3)")
    for p in pitch_angles:
        _angle(p, "pitch angle")
    _angle(fov_deg, "FOV")
    return _exact_maps.pitch_map_stack(output_width, output_height, list(pitch_angles), int(shape[1]), int(shape[0]), fov_deg)


def process_views(pano_image, yaw_angles, pitch_angles, output_width, output_height, fov_deg=90, device=None,
                  pixel_path=None, maps=None, exact=None):
    """All yaws x pitches of one panorama in one kernel launch.
    Returns uint8 [n_yaw][n_pitch][output_height][output_width][3]; the array lives in page-locked host
    memory (pooled; P2P_PINNED=0 for ordinary memory) so that the copy back from the GPU is one DMA.
    pixel_path: "u8" = the reference's arithmetic (two fixed-point cv2.remap stages, the default and the only
    parity mode); "f32" / "f16" = the opt-in single float resample with wrap-around (not in the reference);
    None = what set_quality() chose.
    exact: True = the identical-results mode (see the module docstring): host-evaluated pitch maps, device yaw tables,
    every pixel on the GPU; None = what set_exact() chose.
    maps: (yaw_rows, U, V) = the caller's OWN maps, INTEGRATION.md "Option C": yaw_rows [n_yaw][pano_width] =
    row 0 of every get_yaw_mapping()[0] (P:42-52), U / V [n_pitch][H][W] = get_pitch_mapping() outputs (P:55-73) --
    given the reference's own maps the views are the reference's bytes."""
    pixel_path = _QUALITY if pixel_path is None else pixel_path
    exact = (_EXACT and maps is None and pixel_path == "u8") if exact is None else bool(exact)
    dev = _DEVICE if device is None else device
    if maps is not None:
        if pixel_path != "u8":
            raise ValueError("caller maps go through the fixed-point path only")
        yaw_rows, U, V = maps
        return _native.remap_views_maps(pano_image, yaw_rows, U, V, dev)
    yaws = [_angle(y, "yaw angle") for y in yaw_angles]
    if exact:
        if pixel_path != "u8":
            raise ValueError("exact=True is the reference's fixed-point arithmetic; it has no float pixel path")
        U, V, key = _exact_maps_for(pano_image, pitch_angles, output_width, output_height, fov_deg)
        return _native.remap_views_pitch_maps(pano_image, yaws, U, V, key, dev, pinned=_PINNED)
    pitches = [_angle(p, "pitch angle") for p in pitch_angles]
    return _native.remap_views_f64(pano_image, yaws, pitches, _angle(fov_deg, "FOV"),
                                   output_width, output_height, dev,
                                   pinned=_PINNED, flags=_flags_of(pixel_path))


def process_yaw_and_pitchs(pano_image, yaw_angle, pitch_angles, output_width, output_height, fov_deg=90, *, exact=None):
    """Drop-in for P:181-221: one yaw, several pitches -> list of (output_height, output_width, 3) uint8.
    exact (keyword-only, additive): True = the reference's bytes (host-evaluated pitch maps); None = set_exact()'s choice."""
    logging.debug(f"[Yaw/Pitch] Starting processing for yaw_angle={yaw_angle}")
    views = process_views(pano_image, [yaw_angle], list(pitch_angles), output_width, output_height, fov_deg, exact=exact)
    return [views[0, i] for i in range(views.shape[1])]


# ----------------------------------------------------------------------------------------------
# image files (Pillow, presented with cv2.imread / cv2.imwrite conventions)
# ----------------------------------------------------------------------------------------------
# Per-stage host time of the file-to-file path, for tools/cli_end_to_end.py: None (nothing is recorded) or a dict that
# collects thread-seconds per stage -- "decode" (file -> RGB array), "to_pinned" (the one copy into page-locked memory;
# no channel swap), "device_wait" (the writer side waiting for an image's download), "encode" (array -> PNG / JPEG
# bytes), "write" (bytes -> file).
stage_seconds = None
_stage_lock = __import__("threading").Lock()


def _stage(name, t0):
    if stage_seconds is not None:
        dt = __import__("time").perf_counter() - t0
        with _stage_lock:
            stage_seconds[name] = stage_seconds.get(name, 0.0) + dt


def _now():
    return __import__("time").perf_counter() if stage_seconds is not None else 0.0


def _decode_rgb(path):
    """Decode an image file to an RGB HxWx3 uint8 array view (Pillow's memory); raises on any failure."""
    from PIL import Image, ImageOps

    t0 = _now()
    with Image.open(str(path)) as im:
        im = ImageOps.exif_transpose(im)
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):
            arr = (np.asarray(im, dtype=np.uint32) >> 8).astype(np.uint8)
            out = np.stack([arr, arr, arr], axis=-1)
        else:
            out = np.asarray(im.convert("RGB"), dtype=np.uint8)
    _stage("decode", t0)
    return out


def _to_pinned(arr):
    """A contiguous copy of `arr` (any channel order / strides) in page-locked memory, from which the upload is one
    DMA; ordinary memory when no device / P2P_PINNED=0 (the compute call reports that)."""
    t0 = _now()
    try:
        if _PINNED:
            try:
                dst = _native.pinned_empty(arr.shape)
                dst[...] = arr
                return dst
            except (_native.P2PError, OSError, MemoryError):
                pass
        return np.ascontiguousarray(arr)
    finally:
        _stage("to_pinned", t0)


def _imread_rgb(path):
    """What the file-to-file path decodes with: HxWx3 uint8 in Pillow's RGB order, page-locked, or None when the file
    cannot be decoded (P:244-247).  The kernels do not care which channel is which and nobody else sees the array,
    so no channel is swapped between the decoder and the encoder (cv2.imread's BGR is an artefact of cv2, P:243-244)."""
    try:
        return _to_pinned(_decode_rgb(path))
    except Exception:
        return None


def _imread_bgr(path):
    """cv2.imread(path) stand-in for callers that are handed the array: HxWx3 uint8 BGR, or None (P:244-247)."""
    try:
        return _to_pinned(_decode_rgb(path)[:, :, ::-1])
    except Exception:
        return None


def _imwrite_rgb(path, image):
    """Encode an RGB HxWx3 uint8 array as .png / .jpg / .jpeg with cv2.imwrite's defaults (P:277): JPEG quality 95; PNG
    rows filtered with SUB and deflated at Z_BEST_SPEED / Z_RLE (_png.py: OpenCV's own settings, 2.4 x faster than
    Pillow's encoder at its fastest level -- and the encoder is what the tool waits for)."""
    import io

    t0 = _now()
    if Path(path).suffix.lower() in (".jpg", ".jpeg"):
        from PIL import Image

        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(image)).save(buf, format="JPEG", quality=95)
        data = buf.getbuffer()
    else:
        try:
            data = _png.encode_png(image)
        except ValueError:  # (not a uint8 image of 1 to 4 channels: Pillow's general encoder)
            from PIL import Image

            buf = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(image)).save(buf, format="PNG", compress_level=1)
            data = buf.getbuffer()
    _stage("encode", t0)
    t0 = _now()
    with open(str(path), "wb") as f:
        f.write(data)
    _stage("write", t0)
    return True


def _imwrite_bgr(path, image):
    """cv2.imwrite(path, image) stand-in for arrays in cv2's BGR order."""
    return _imwrite_rgb(path, image[:, :, ::-1])


# throughput summary of a main() run (keep the reference's log format, add one Mpix/s + GB/s line)
_stats_lock = __import__("threading").Lock()
_stats = {"views": 0, "pixels": 0, "src_bytes": 0}


def _count(views=0, pixels=0, src_bytes=0):
    with _stats_lock:
        _stats["views"] += views
        _stats["pixels"] += pixels
        _stats["src_bytes"] += src_bytes


def _write_view(view, yaw_angle, pitch_angle, base_name, output_width, output_height, output_format, output_dir):
    """Encode and write ONE view (one task of the writer pool).  The reference's parallel unit is the yaw (P:252-265) and
    its files are written by the caller, one after the other (P:271-278); with the resampling on the GPU the encoder is what
    the tool waits for, and a yaw's pitch views are independent files: a task each -- 20 tasks per image of the
    reference's default view set instead of 4, which is what keeps more than a handful of workers busy."""
    _count(views=1, pixels=int(output_width) * int(output_height))
    out_filename = f"{base_name}_{output_width}x{output_height}_yaw_{yaw_angle}_pitch_{pitch_angle}.{output_format}"
    output_file = output_dir / out_filename
    _imwrite_rgb(output_file, view)  # (the image was decoded to RGB and never swapped: _imread_rgb)
    logging.debug(f"Saved {output_file}")


def process_single_image(
    input_image_path,
    output_dir,
    yaw_angles,
    pitch_angles,
    output_width,
    output_height,
    num_workers=4,
    output_format="png",
    fov_deg=90,
):
    """Drop-in for P:227-280.  The reference fans one task per yaw out to `num_workers` threads that
    each resample on the CPU and the caller writes the files; here every yaw and pitch of the image
    comes from ONE kernel launch and the `num_workers` threads encode / write the files, one task
    per view (the codecs release the GIL), which is where the time goes once the resampling is on the GPU."""
    logging.info(f"Loading image: {input_image_path}")
    input_image = _imread_rgb(input_image_path)
    if input_image is None:
        logging.error(f"Failed to read image: {input_image_path}")
        return
    _process_decoded_image(input_image, input_image_path, output_dir, yaw_angles, pitch_angles, output_width,
                           output_height, num_workers, output_format, fov_deg)


def _mode_of(input_image, pitch_angles, output_width, output_height, fov_deg):
    """(flags, maps) of the module's current mode for one image: maps = (U, V, key) in exact mode, else None."""
    if _EXACT:
        return 0, _exact_maps_for(input_image, pitch_angles, output_width, output_height, fov_deg)
    return _flags_of(_QUALITY), None


def _views_of(input_image, yaw_angles, pitch_angles, output_width, output_height, fov_deg, device=None):
    """Every view of one image.  With several devices selected (set_devices) and this single image to draw, the
    image is shared out to them -- a band of rows of every view each, or runs of the view list;
    otherwise one device."""
    if device is None and _DEVICES and len(_DEVICES) > 1:
        flags, maps = _mode_of(input_image, pitch_angles, output_width, output_height, fov_deg)
        return _driver.process_views_sharded(input_image, [_angle(y, "yaw angle") for y in yaw_angles],
                                             [_angle(p, "pitch angle") for p in pitch_angles], output_width,
                                             output_height, _angle(fov_deg, "FOV"), _DEVICES, flags=flags, maps=maps)
    if device is None:
        return process_views(input_image, yaw_angles, pitch_angles, output_width, output_height, fov_deg)
    return process_views(input_image, yaw_angles, pitch_angles, output_width, output_height, fov_deg, device=device)


class _SyncPipeline:
    """The folder walk's device stage without overlap: one synchronous call per image (P2P_PIPELINE=0, and what the
    host-logic tests substitute)."""

    slots = (None,)

    def __init__(self, device=None):
        self.device = device

    def submit(self, pano, yaws, pitches, fov, ow, oh, flags=0, maps=None):
        if self.device is None:
            return process_views(pano, yaws, pitches, ow, oh, fov)
        return process_views(pano, yaws, pitches, ow, oh, fov, device=self.device)

    def close(self):
        pass


def _make_pipeline(device):
    if os.environ.get("P2P_PIPELINE", "1") == "0":
        return _SyncPipeline(device)
    return _driver.DevicePipeline(_DEVICE if device is None else device)


def _submit_writes(executor, views, input_image_path, output_dir, yaw_angles, pitch_angles, output_width,
                   output_height, output_format):
    """One write task per VIEW on `executor`, grouped by yaw (the reference's parallel unit and the unit its errors are
    reported by, P:252-280).  `views` is the array, a ticket of the device pipeline (its download may still be in flight),
    or the exception that replaced them.  Returns [(yaw_angle, [futures] or exception)], to be drained with _drain_views."""
    base_name = Path(input_image_path).stem
    output_dir = Path(output_dir)
    yaw_angles = list(yaw_angles)
    try:
        if isinstance(views, Exception):
            raise views
        if hasattr(views, "result"):
            t0 = _now()
            views = views.result()
            _stage("device_wait", t0)
    except Exception as e:  # the reference reports task failures per yaw and carries on (P:279-280)
        return [(yaw_angle, e) for yaw_angle in yaw_angles]
    return [(yaw_angle, [executor.submit(_write_view, views[yi][pi], yaw_angle, pitch_angle, base_name, output_width,
                                         output_height, output_format, output_dir)
                         for pi, pitch_angle in enumerate(pitch_angles)])
            for yi, yaw_angle in enumerate(yaw_angles)]


def _submit_views(executor, input_image, input_image_path, output_dir, yaw_angles, pitch_angles, output_width,
                  output_height, output_format, fov_deg, device=None):
    """One kernel launch for every yaw and pitch of the image, then one write task per view on `executor`."""
    yaw_angles = list(yaw_angles)
    _count(src_bytes=int(getattr(input_image, "nbytes", 0)))
    try:
        views = _views_of(input_image, yaw_angles, pitch_angles, output_width, output_height, fov_deg, device)
    except Exception as e:
        views = e
    return _submit_writes(executor, views, input_image_path, output_dir, yaw_angles, pitch_angles, output_width,
                          output_height, output_format)


def _drain_views(tasks):
    """Wait for an image's write tasks; a failed yaw is logged and the others go on (P:271-280)."""
    from tqdm import tqdm

    for yaw_angle, task in tqdm(tasks, desc="Processing yaw angles"):
        try:
            if isinstance(task, Exception):
                raise task
            first_error = None
            for view_task in task:  # (every file of the yaw is waited for; the first failure is the yaw's)
                try:
                    view_task.result()
                except Exception as e:
                    first_error = first_error or e
            if first_error is not None:
                raise first_error
        except Exception as e:
            logging.error(f"Error processing yaw_angle {yaw_angle}: {e}")


def _process_decoded_image(input_image, input_image_path, output_dir, yaw_angles, pitch_angles, output_width,
                           output_height, num_workers, output_format, fov_deg, device=None):
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=max(1, int(num_workers or 1))) as executor:
        _drain_views(_submit_views(executor, input_image, input_image_path, output_dir, yaw_angles, pitch_angles,
                                   output_width, output_height, output_format, fov_deg, device))


def main(
    input_path,
    output_path,
    yaw_angles,
    pitch_angles,
    output_width,
    output_height,
    num_workers=None,
    output_format="png",
    fov_deg=90,
    enable_file_logging=False,
):
    """Drop-in for P:286-356: one image or every .jpg/.jpeg/.png under a directory (recursive)."""
    import time as _time

    t_start = _time.perf_counter()
    with _stats_lock:
        _stats.update(views=0, pixels=0, src_bytes=0)
    if num_workers is None:
        cpu_cores = os.cpu_count() or 1
        num_workers = max(1, int(cpu_cores * 0.9))
        logging.info(f"No num_workers specified. Using {num_workers} (~90% of CPU cores).")
    else:
        logging.info(f"Using {num_workers} worker threads.")

    output_dir = Path(output_path)
    output_dir.mkdir(parents=True, exist_ok=True)
    logging.info(f"Output directory set to: {output_dir}")

    input_path_obj = Path(input_path)
    common = dict(
        output_dir=output_dir,
        yaw_angles=yaw_angles,
        pitch_angles=pitch_angles,
        output_width=output_width,
        output_height=output_height,
        num_workers=num_workers,
        output_format=output_format,
        fov_deg=fov_deg,
    )
    if input_path_obj.is_dir():
        valid_exts = {".jpg", ".jpeg", ".png"}
        all_images = [f for f in input_path_obj.rglob("*") if f.suffix.lower() in valid_exts]
        if not all_images:
            logging.warning(f"No images found in directory: {input_path_obj}")
            return
        logging.info(f"Found {len(all_images)} images in folder: {input_path_obj}")
        from concurrent.futures import ThreadPoolExecutor

        def run_share(image_files, device):
            # The reference walks the images one after the other (P:330-341) with the yaws of one image in
            # parallel.  Same order of work here, as a pipeline with every stage overlapped: files are decoded a
            # few ahead on helper threads; the device keeps two images in flight (upload of image k+1 and download
            # of image k-1 under the kernel of image k: _driver.DevicePipeline); the files of finished images are
            # encoded by the shared writer pool, one task per yaw.
            from collections import deque

            # Files decoded ahead: 3, more with many workers -- an 8K PNG takes one thread 0.6 s, the GPU 0.1 ms, and with 16
            # workers the 20 default views of an image are encoded in 0.35 s: three decoders were what the tool waited for
            # (tools/cli_end_to_end.py).  Each is a page-locked panorama (100 MB at 8K).
            depth = max(3, min(8, int(num_workers or 1) // 2))
            n_dec = max(1, min(depth, int(num_workers or 1)))
            yaws = [_angle(y, "yaw angle") for y in yaw_angles]
            pitches = [_angle(p, "pitch angle") for p in pitch_angles]
            pipe = _make_pipeline(device)
            try:
                with ThreadPoolExecutor(max_workers=n_dec) as decoder, \
                        ThreadPoolExecutor(max_workers=max(1, int(num_workers or 1))) as writers:
                    decoding = deque(decoder.submit(_imread_rgb, f) for f in image_files[:depth])
                    on_device, writing = deque(), deque()

                    def retire(n_keep):
                        while len(on_device) > n_keep:
                            ticket, image_file = on_device.popleft()
                            writing.append(_submit_writes(writers, ticket, image_file, output_dir, yaw_angles,
                                                          pitch_angles, output_width, output_height, output_format))
                            while len(writing) >= depth:
                                _drain_views(writing.popleft())

                    for k, image_file in enumerate(image_files):
                        decoded = decoding.popleft().result()
                        if k + depth < len(image_files):
                            decoding.append(decoder.submit(_imread_rgb, image_files[k + depth]))
                        logging.info(f"Loading image: {image_file}")
                        if decoded is None:
                            logging.error(f"Failed to read image: {image_file}")
                            continue
                        _count(src_bytes=int(decoded.nbytes))
                        try:
                            flags, maps = _mode_of(decoded, pitch_angles, output_width, output_height, fov_deg)
                            ticket = pipe.submit(decoded, yaws, pitches, _angle(fov_deg, "FOV"), output_width, output_height,
                                                 flags=flags, maps=maps)
                        except Exception as e:
                            ticket = e
                        on_device.append((ticket, image_file))
                        del decoded
                        retire(len(pipe.slots) - 1)
                    retire(0)
                    while writing:
                        _drain_views(writing.popleft())
            finally:
                pipe.close()

        if _DEVICES and len(_DEVICES) > 1 and len(all_images) < len(_DEVICES):
            # fewer images than GPUs: every image is drawn by all of them, each its run of the view list
            logging.info(f"Dealing the views of each of {len(all_images)} images to devices {_DEVICES}")
            for image_file in all_images:
                process_single_image(input_image_path=image_file, **common)
        elif _DEVICES and len(_DEVICES) > 1:
            # one host thread per GPU, images dealt round-robin, nothing exchanged between devices
            shares = [(all_images[i::len(_DEVICES)], d) for i, d in enumerate(_DEVICES)]
            shares = [(files, d) for files, d in shares if files]
            logging.info(f"Dealing {len(all_images)} images round-robin to devices {[d for _, d in shares]}")
            with ThreadPoolExecutor(max_workers=len(shares)) as per_device:
                for f in [per_device.submit(run_share, files, d) for files, d in shares]:
                    f.result()
        else:
            run_share(all_images, _DEVICES[0] if _DEVICES else None)
    else:
        process_single_image(input_image_path=input_path_obj, **common)

    logging.info("All processing completed.")
    dt = max(_time.perf_counter() - t_start, 1e-9)
    with _stats_lock:
        n_views, n_px, n_src = _stats["views"], _stats["pixels"], _stats["src_bytes"]
    if n_views:
        logging.info(f"{n_views} views, {n_px / 1e6:.1f} Mpix in {dt:.2f} s: {n_px / 1e6 / dt:.1f} Mpix/s end to end "
                     f"(decode and encode included), {(n_src + 3 * n_px) / 1e9 / dt:.2f} GB/s of pixels through the GPU")


def check_pitch(value: str) -> int:
    """Pitch validator of P:362-376 (integer, 1..179)."""
    try:
        pitch = int(value)
    except ValueError:
        raise argparse.ArgumentTypeError(f"Pitch angle must be an integer, got '{value}'.")
    if not (1 <= pitch <= 179):
        raise argparse.ArgumentTypeError(f"Pitch angle must be between 1 and 179 degrees, got {pitch}.")
    return pitch


def build_arg_parser():
    """The reference's argparse surface (P:383-455), flag for flag, plus --device."""
    p = argparse.ArgumentParser(
        description="Process panorama images or an entire folder of images into planar projections."
    )
    p.add_argument("--input_path", type=str, required=True,
                   help="Path to the input panorama image or folder of images")
    p.add_argument("--output_path", type=str, default="output_images", help="Path to save the output images")
    p.add_argument("--output_format", type=str, choices=["png", "jpg", "jpeg"], default="png",
                   help="Output image format (png, jpg, jpeg)")
    p.add_argument("--FOV", type=int, default=90, help="Field of View in degrees")
    p.add_argument("--output_width", type=int, default=800, help="Width of the output image in pixels")
    p.add_argument("--output_height", type=int, default=800, help="Height of the output image in pixels")
    p.add_argument("--pitch_angles", nargs="+", type=check_pitch, default=[30, 60, 90, 120, 150],
                   help="List of pitch angles in degrees (1-179). e.g. --pitch_angles 30 60 90")
    p.add_argument("--yaw_angles", nargs="+", type=int, default=[0, 90, 180, 270],
                   help="List of yaw angles in degrees (0-360). e.g. --yaw_angles 0 90 180 270")
    p.add_argument("--num_workers", type=int, default=None,
                   help="Number of worker threads for parallel yaw processing. If not specified, uses ~90%% of CPU cores.")
    p.add_argument("--enable_file_logging", action="store_true", help="Enable logging to a file.")
    p.add_argument("-v", "--version", action="version", version=f"%(prog)s {get_version()}",
                   help="Show version information")
    p.add_argument("--device", type=int, default=None, help="HIP device index (default 0 or $P2P_DEVICE)")
    p.add_argument("--devices", type=int, nargs="+", default=None,
                   help="HIP devices a folder of images is dealt to round-robin (one host thread per device)")
    p.add_argument("--exact", action="store_true",
                   help="Identical-results mode: the pitch maps are evaluated on the host exactly as the reference does "
                        "(NumPy float32, P:114-175) and every pixel is drawn from them on the GPU -- the reference's bytes "
                        "on any panorama.  Default: maps on the device too (+-1 level on smooth images)")
    p.add_argument("--quality", choices=sorted(_PIXEL_PATHS), default="u8",
                   help="Pixel arithmetic: u8 = the reference's two fixed-point cv2.remap stages (default); f32 / f16 = one "
                        "float resample per view with true wrap-around at the seam (not in the reference)")
    p.add_argument("--pixel_centres", action="store_true",
                   help="With --quality f32 / f16 only: sample through pixel centres (the reference samples at integer "
                        "coordinates, which shifts the picture by half a pixel)")
    return p


def cli(argv=None):
    parser = build_arg_parser()
    args = parser.parse_args(argv)
    if args.exact and args.quality != "u8":
        parser.error("--exact is the reference's fixed-point arithmetic: it goes with --quality u8 only")
    if args.pixel_centres and args.quality == "u8":
        parser.error("--pixel_centres goes with --quality f32 / f16 (u8 is the reference's arithmetic, integer coordinates)")
    # logging set-up as P:462-475 (the logs/ directory is created even without file logging)
    log_file_path = Path(__file__).resolve().parent.parent / "logs" / "app.log"
    log_file_path.parent.mkdir(parents=True, exist_ok=True)
    handlers = [logging.StreamHandler()]
    if args.enable_file_logging:
        handlers.append(logging.FileHandler(log_file_path, mode="a"))
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)s] %(message)s", handlers=handlers)
    if args.device is not None:
        set_device(args.device)
    set_devices(args.devices)
    set_exact(False)
    set_quality(args.quality, args.pixel_centres)
    set_exact(args.exact)
    main(
        input_path=args.input_path,
        output_path=args.output_path,
        yaw_angles=args.yaw_angles,
        pitch_angles=args.pitch_angles,
        output_width=args.output_width,
        output_height=args.output_height,
        num_workers=args.num_workers,
        output_format=args.output_format,
        fov_deg=args.FOV,
        enable_file_logging=args.enable_file_logging,
    )


if __name__ == "__main__":
    cli()
