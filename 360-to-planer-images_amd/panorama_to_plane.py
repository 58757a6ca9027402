"""Drop-in for the reference's legacy tool, app/legacy/panorama_to_plane.py ("L"): one combined
yaw + pitch rotation per view and a single cv2.remap with BORDER_REFLECT.

    reference                                    L:line     here
    get_rotation_matrix(yaw_rad, pitch_rad)      L:21-45    float32 R_pitch @ R_yaw (nine host scalars, as NumPy builds them)
    precompute_mapping(W, H, FOV_rad, yaw, ...)  L:47-157   p2p_build_rot_map (rot_map_kernel), lru_cache'd like L:47
    interpolate_color(U, V, img, method)         L:159-180  p2p_remap_maps_interp_u8 (nearest / bilinear / bicubic, BORDER_REFLECT)
    panorama_to_plane(pano_array, U, V)          L:182-194  interpolate_color(U, V, pano_array)
    the command line                             L:196-388  same flags, defaults, value ranges and file names -- the DRIVER is
                                                            this build's own: images decoded ahead on a thread pool, one upload
                                                            and ONE launch per image for all its yaws, encoders on a second pool

--exact (additive; set_exact(True) from Python): the identical-results mode of this package -- precompute_mapping is
evaluated on the host exactly as the reference evaluates it (_exact_maps.legacy_mapping follows L:47-157: NumPy float32
flow, libm, one sgemm) and every pixel is drawn from those maps on the GPU: the reference's bytes.  Default: the maps
come from the device too (rot_map_kernel, within 1e-5 of the host's).

Image files go through Pillow (cv2 is not a dependency here).  The reference converts BGR -> RGB after
imread and back before imwrite (L:254, L:275); remap is channel-agnostic, so the files are the same.
There is no CPU fallback for the maps or the resampling.
"""
import argparse
import logging
import os
import sys
from concurrent.futures import ThreadPoolExecutor
from functools import lru_cache
from pathlib import Path
from typing import Iterable, Tuple

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _exact_maps  # type: ignore
    import _native  # type: ignore
    import _png  # type: ignore
else:
    from . import _exact_maps, _native, _png

_DEVICE = int(os.environ.get("P2P_DEVICE", "0"))
_EXACT = False  # set_exact / --exact


def set_exact(on=True):
    """Maps as the reference computes them, on the host (see the module docstring); pixels on the GPU either way."""
    global _EXACT
    if bool(on) != _EXACT:
        precompute_mapping.cache_clear()  # (the cached maps belong to the other mode)
    _EXACT = bool(on)


def get_rotation_matrix(yaw_radian: float, pitch_radian: float) -> np.ndarray:
    """L:21-45.  Two float32 3x3 arrays multiplied with np.dot: nine scalars, evaluated on the host as the
    reference does."""
    R_yaw = np.array(
        [[np.cos(yaw_radian), 0, np.sin(yaw_radian)], [0, 1, 0], [-np.sin(yaw_radian), 0, np.cos(yaw_radian)]],
        dtype=np.float32,
    )
    R_pitch = np.array(
        [[1, 0, 0], [0, np.cos(pitch_radian), -np.sin(pitch_radian)], [0, np.sin(pitch_radian), np.cos(pitch_radian)]],
        dtype=np.float32,
    )
    return np.dot(R_pitch, R_yaw)


@lru_cache(maxsize=None)
def precompute_mapping(W: int, H: int, FOV_rad: float, yaw_radian: float, pitch_radian: float,
                       pano_width: int, pano_height: int) -> Tuple[np.ndarray, np.ndarray]:
    """L:47-157: (U, V) float32 maps of one view, computed by rot_map_kernel -- in exact mode on the host, with the
    reference's own arithmetic."""
    if _EXACT:
        return _exact_maps.legacy_mapping(W, H, FOV_rad, yaw_radian, pitch_radian, pano_width, pano_height)
    R = get_rotation_matrix(yaw_radian, pitch_radian)
    return _native.build_rot_map(W, H, float(FOV_rad), R, pano_width, pano_height, _DEVICE)


_METHODS = {"nearest": _native.INTER_NEAREST, "bilinear": _native.INTER_LINEAR, "bicubic": _native.INTER_CUBIC}


def interpolate_color(U: np.ndarray, V: np.ndarray, img: np.ndarray, method: str = "bilinear") -> np.ndarray:
    """L:159-180.  An unknown method name falls back to bilinear, as the reference's dict.get does (L:177)."""
    interp = _METHODS.get(method, _native.INTER_LINEAR)
    return _native.remap_maps(img, U, V, border=_native.BORDER_REFLECT, device=_DEVICE, interpolation=interp)


def panorama_to_plane(pano_array: np.ndarray, U: np.ndarray, V: np.ndarray) -> np.ndarray:
    return interpolate_color(U, V, pano_array)


# ------------------------------------------------------------------------------------------------
# Command line.  SURVEY 8(f)3 asks for the maps and the interpolation methods above; the folder tool around them keeps the
# reference's flags, defaults, value ranges and output names (L:196-237, L:268, L:285-304) and is otherwise built like
# this package's current tool (panorama_to_plane_pitch.py): decode ahead -> one device stage -> encode behind.
# ------------------------------------------------------------------------------------------------
_SUFFIXES = (".jpg", ".jpeg", ".png")  # the patterns the reference globs for (L:325)

# flag -> argparse keywords, the reference's table (L:285-299)
_FLAGS = (
    ("--input_path", dict(type=str, required=True, help="Path to the input panorama images")),
    ("--output_path", dict(type=str, default="output_images", help="Path to save the output images")),
    ("--output_format", dict(type=str, choices=["png", "jpg", "jpeg"], help="Output image format (png, jpg, jpeg)")),
    ("--FOV", dict(type=int, default=90, help="Field of View in degrees")),
    ("--output_width", dict(type=int, default=1000, help="Width of the output image in pixels")),
    ("--output_height", dict(type=int, default=1500, help="Height of the output image in pixels")),
    ("--pitch", dict(default=90, help="Pitch angle in degrees (1-179)")),
    ("--yaw_angles", dict(nargs="+", type=int, default=[0, 60, 120, 180, 240, 300],
                          help="List of yaw angles in degrees (0-360). Example: --yaw_angles 0 60 120 180 240 300")),
    ("--num_workers", dict(type=int, default=None,
                           help="Number of worker threads. Defaults to 90%% of CPU cores if not specified.")),
)


def check_pitch(value) -> int:
    """argparse type of --pitch: an integer in 1..179 (the range and the two messages of L:196-216)."""
    try:
        pitch = int(value)
    except (TypeError, ValueError):
        raise argparse.ArgumentTypeError("Pitch value must be an integer between 1 and 179.") from None
    if pitch < 1 or pitch > 179:
        raise argparse.ArgumentTypeError(f"{pitch} is an invalid pitch value. It must be between 1 and 179.")
    return pitch


def check_yaw(yaw_angles: Iterable[int]) -> list:
    """The yaw list as the reference uses it (L:218-237): every value in 0..360, each once, ascending."""
    wanted = sorted(set(yaw_angles))
    for yaw in wanted:
        if yaw < 0 or yaw > 360:
            raise argparse.ArgumentTypeError(f"{yaw} is an invalid yaw value. It must be between 0 and 360.")
    return wanted


def build_arg_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Convert panorama images to plane projections based on FOV, yaw, and pitch.")
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **(dict(kw, type=check_pitch) if flag == "--pitch" else kw))
    parser.add_argument("--exact", action="store_true",
                        help="Identical-results mode: the maps are evaluated on the host exactly as the reference does "
                             "(NumPy float32, L:47-157); every pixel is still drawn on the GPU")
    return parser


def parse_arguments(argv=None) -> argparse.Namespace:
    args = build_arg_parser().parse_args(argv)
    args.yaw_angles = check_yaw(args.yaw_angles)
    return args


def _decode_rgb(path):
    """The file as an RGB uint8 array (what cv2.imread + COLOR_BGR2RGB give the reference, L:249-254), or None."""
    try:
        from PIL import Image, ImageOps

        with Image.open(str(path)) as im:
            return np.ascontiguousarray(np.asarray(ImageOps.exif_transpose(im).convert("RGB"), dtype=np.uint8))
    except Exception:
        return None


def _encode_rgb(path, image):
    """cv2.imwrite's defaults (L:275-278): JPEG quality 95; PNG rows filtered with SUB, deflated at Z_BEST_SPEED / Z_RLE
    (_png.py)."""
    if Path(path).suffix.lower() in (".jpg", ".jpeg"):
        from PIL import Image

        Image.fromarray(image).save(str(path), format="JPEG", quality=95)
    else:
        try:
            data = _png.encode_png(image)
        except ValueError:  # (not a uint8 image of 1 to 4 channels: Pillow's general encoder)
            from PIL import Image

            Image.fromarray(image).save(str(path), format="PNG", compress_level=1)
        else:
            with open(str(path), "wb") as f:
                f.write(data)
    logging.info(f"Saved output image to {path}")


def views_of_image(pano, maps_by_yaw):
    """Every yaw's view of one image: the image goes to the device once and all maps are drawn by one launch
    (p2p_remap_maps_batch_u8; the reference calls panorama_to_plane once per yaw, L:259-265 -- the same pixels).
    Anything but a 3-channel image takes the one-map entry point."""
    yaws = list(maps_by_yaw)
    if not yaws:
        return {}
    if pano.ndim == 3 and pano.shape[2] == 3:
        batch = _native.remap_maps_batch(pano, np.stack([maps_by_yaw[y][0] for y in yaws]),
                                         np.stack([maps_by_yaw[y][1] for y in yaws]), border=_native.BORDER_REFLECT, device=_DEVICE)
        return dict(zip(yaws, batch))
    return {y: panorama_to_plane(pano, *maps_by_yaw[y]) for y in yaws}


class ResidentViews:
    """Every yaw's view of MANY images of one size: a resident job (p2p_job_set_border + p2p_job_set_maps) that keeps the
    maps and their plan on the device -- the reference builds its maps once per run too (L:341-363) and then calls
    panorama_to_plane once per image and yaw (L:259-265).  Per image: one upload, one launch for all yaws, one download."""

    def __init__(self, size, maps_by_yaw, device=None):
        self.yaws = list(maps_by_yaw)
        U = np.stack([maps_by_yaw[y][0] for y in self.yaws])
        V = np.stack([maps_by_yaw[y][1] for y in self.yaws])
        self.ctx = _native.Context(_DEVICE if device is None else device)
        try:
            # ONE yaw of 0 degrees (the yaw stage is then a copy) and the caller's maps as the job's "pitch views"
            self.job = _native.Job(self.ctx, size[1], size[0], 1, [0.0], [90.0] * len(self.yaws), 90.0, U.shape[2], U.shape[1])
            self.job.set_border(_native.BORDER_REFLECT)
            self.job.set_maps(None, U, V)
        except Exception:
            self.ctx.close()
            raise

    def views(self, pano):
        self.job.set_pano(0, pano)
        self.job.run()
        return dict(zip(self.yaws, self.job.get_views(0, pinned=True)[0]))

    def close(self):
        self.job.close()
        self.ctx.close()


def convert_folder(input_path, output_path, yaw_angles, pitch=90, FOV=90, output_width=1000, output_height=1500,
                   output_format=None, num_workers=None) -> int:
    """The legacy tool's job (L:306-388) for one folder; returns the number of files written.  Images are decoded
    ahead of the device on `num_workers` threads and encoded behind it on as many; the device stage itself -- one upload
    and one launch per image -- runs on the calling thread.  A file that cannot be read, or a view that cannot be
    written, is logged and skipped, as the reference's per-image try / except does (L:282-283)."""
    src, dst = Path(input_path), Path(output_path)
    if not src.is_dir():
        logging.error(f"Input path {src} is not a directory or does not exist.")
        return 0
    files = sorted(p for p in src.iterdir() if p.is_file() and p.suffix in _SUFFIXES)
    if not files:
        logging.warning(f"No images found in {src} with extensions {['*' + e for e in _SUFFIXES]}.")
        return 0
    dst.mkdir(parents=True, exist_ok=True)
    workers = num_workers if num_workers else max(1, int((os.cpu_count() or 1) * 0.9))
    logging.info(f"Using {workers} worker(s) for processing.")
    logging.info(f"Starting processing of {len(files)} images with {len(yaw_angles)} yaw angles each.")
    written, maps_for, resident = [], {}, {}
    with ThreadPoolExecutor(max_workers=workers) as readers, ThreadPoolExecutor(max_workers=workers) as writers:
        decoded = [(f, readers.submit(_decode_rgb, f)) for f in files]
        try:
            from tqdm import tqdm

            decoded = tqdm(decoded, desc="Processing images", unit="image")
        except ImportError:
            pass
        for f, fut in decoded:
            pano = fut.result()
            if pano is None:
                logging.error(f"Failed to read image {f}. Skipping.")
                continue
            logging.info(f"Processing {f}...")
            try:
                # the maps belong to a panorama SIZE (the reference builds them once, from its first image, L:341-363)
                size = pano.shape[:2]
                if size not in maps_for:
                    maps_for[size] = {y: precompute_mapping(W=output_width, H=output_height, FOV_rad=float(np.radians(FOV)),
                                                            yaw_radian=float(np.radians(y)), pitch_radian=float(np.radians(pitch)),
                                                            pano_width=size[1], pano_height=size[0]) for y in yaw_angles}
                fmt = output_format or f.suffix[1:]
                if pano.ndim == 3 and pano.shape[2] == 3 and maps_for[size]:
                    if size not in resident:
                        resident[size] = ResidentViews(size, maps_for[size])
                    views = resident[size].views(pano)
                else:
                    views = views_of_image(pano, maps_for[size])
                for yaw, view in views.items():
                    written.append(writers.submit(_encode_rgb, dst / f"{f.stem}_pitch{pitch}_yaw{yaw}_fov{FOV}.{fmt}", view))
            except Exception as e:
                logging.error(f"Failed to process {f}: {e}")
        for r in resident.values():
            r.close()
        done = 0
        for w in written:
            try:
                w.result()
                done += 1
            except Exception as e:
                logging.error(f"Failed to write a view: {e}")
    logging.info("Processing completed.")
    return done


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)s] %(message)s", handlers=[logging.StreamHandler()])
    a = parse_arguments(argv)
    set_exact(a.exact)
    convert_folder(a.input_path, a.output_path, a.yaw_angles, a.pitch, a.FOV, a.output_width, a.output_height,
                   a.output_format, a.num_workers)


if __name__ == "__main__":
    main()
