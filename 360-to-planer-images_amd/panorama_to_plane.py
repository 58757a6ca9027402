"""Drop-in for the reference's legacy tool, app/legacy/panorama_to_plane.py ("L"): one combined
yaw + pitch rotation per view and a single cv2.remap with BORDER_REFLECT.

    reference                                    L:line     here
    get_rotation_matrix(yaw_rad, pitch_rad)      L:21-45    float32 R_pitch @ R_yaw (nine host scalars, as NumPy builds them)
    precompute_mapping(W, H, FOV_rad, yaw, ...)  L:47-157   p2p_build_rot_map (rot_map_kernel), lru_cache'd like L:47
    interpolate_color(U, V, img, method)         L:159-180  p2p_remap_maps_interp_u8 (nearest / bilinear / bicubic, BORDER_REFLECT)
    panorama_to_plane(pano_array, U, V)          L:182-194  interpolate_color(U, V, pano_array)
    check_pitch / check_yaw                      L:196-237  same messages
    process_image_batch / parse_arguments / main L:239-388  same flags, defaults, file names and logging

Image files go through Pillow (cv2 is not a dependency here).  The reference converts BGR -> RGB after
imread and back before imwrite (L:254, L:275); remap is channel-agnostic, so the files are the same.
There is no CPU fallback for the maps or the resampling.
"""
import argparse
import logging
import os
import sys
from concurrent.futures import ThreadPoolExecutor, as_completed
from functools import lru_cache
from pathlib import Path
from typing import List, Tuple

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _native  # type: ignore
else:
    from . import _native

_DEVICE = int(os.environ.get("P2P_DEVICE", "0"))


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
    """L:47-157: (U, V) float32 maps of one view, computed by rot_map_kernel."""
    R = get_rotation_matrix(yaw_radian, pitch_radian)
    return _native.build_rot_map(W, H, float(FOV_rad), R, pano_width, pano_height, _DEVICE)


_METHODS = {"nearest": _native.INTER_NEAREST, "bilinear": _native.INTER_LINEAR, "bicubic": _native.INTER_CUBIC}


def interpolate_color(U: np.ndarray, V: np.ndarray, img: np.ndarray, method: str = "bilinear") -> np.ndarray:
    """L:159-180.  An unknown method name falls back to bilinear, as the reference's dict.get does (L:177)."""
    interp = _METHODS.get(method, _native.INTER_LINEAR)
    return _native.remap_maps(img, U, V, border=_native.BORDER_REFLECT, device=_DEVICE, interpolation=interp)


def panorama_to_plane(pano_array: np.ndarray, U: np.ndarray, V: np.ndarray) -> np.ndarray:
    return interpolate_color(U, V, pano_array)


def check_pitch(value: str) -> int:
    """L:196-216."""
    try:
        ivalue = int(value)
    except ValueError:
        raise argparse.ArgumentTypeError("Pitch value must be an integer between 1 and 179.")
    if not (1 <= ivalue <= 179):
        raise argparse.ArgumentTypeError(f"{ivalue} is an invalid pitch value. It must be between 1 and 179.")
    return ivalue


def check_yaw(yaw_angles: List[int]) -> List[int]:
    """L:218-237: validated, de-duplicated, sorted."""
    unique_yaws = set()
    for val in yaw_angles:
        if not (0 <= val <= 360):
            raise argparse.ArgumentTypeError(f"{val} is an invalid yaw value. It must be between 0 and 360.")
        unique_yaws.add(val)
    return sorted(unique_yaws)


def _imread_rgb(path):
    """cv2.imread + COLOR_BGR2RGB (L:249-254); None when the file cannot be decoded."""
    try:
        from PIL import Image, ImageOps

        with Image.open(str(path)) as im:
            im = ImageOps.exif_transpose(im)
            return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))
    except Exception:
        return None


def _imwrite_rgb(path, image):
    """COLOR_RGB2BGR + cv2.imwrite (L:275-278) with cv2's defaults (JPEG quality 95)."""
    from PIL import Image

    im = Image.fromarray(np.ascontiguousarray(image))
    if Path(path).suffix.lower() in (".jpg", ".jpeg"):
        im.save(str(path), format="JPEG", quality=95)
    else:
        im.save(str(path), format="PNG", compress_level=1)


def process_image_batch(image_path: Path, args: argparse.Namespace, output_path: Path, precomputed_mappings: dict):
    """L:239-283: every yaw of one image; errors are logged and swallowed."""
    logging.info(f"Processing {image_path}...")
    try:
        pano_array = _imread_rgb(image_path)
        if pano_array is None:
            logging.error(f"Failed to read image {image_path}. Skipping.")
            return
        file_name = image_path.stem
        # every yaw of the image in ONE call: the image is uploaded once and all maps are drawn by one launch
        # (the reference calls panorama_to_plane once per yaw, L:259-265; same pixels)
        yaws = list(args.yaw_angles)
        if pano_array.ndim == 3 and pano_array.shape[2] == 3 and yaws:
            views = _native.remap_maps_batch(pano_array, np.stack([precomputed_mappings[y][0] for y in yaws]),
                                             np.stack([precomputed_mappings[y][1] for y in yaws]),
                                             border=_native.BORDER_REFLECT, device=_DEVICE)
        else:
            views = None
        for k, yaw in enumerate(yaws):
            logging.debug(f"Processing {image_path} with yaw {yaw}°...")
            U, V = precomputed_mappings[yaw]
            output_image_array = views[k] if views is not None else panorama_to_plane(pano_array, U, V)
            output_format = args.output_format if args.output_format else image_path.suffix[1:]
            output_image_name = f"{file_name}_pitch{args.pitch}_yaw{yaw}_fov{args.FOV}.{output_format}"
            output_image_path = output_path / output_image_name
            _imwrite_rgb(output_image_path, output_image_array)
            logging.info(f"Saved output image to {output_image_path}")
    except Exception as e:
        logging.error(f"Failed to process {image_path}: {e}")


def build_arg_parser() -> argparse.ArgumentParser:
    """The flags of L:285-304."""
    parser = argparse.ArgumentParser(
        description="Convert panorama images to plane projections based on FOV, yaw, and pitch."
    )
    parser.add_argument("--input_path", type=str, help="Path to the input panorama images", required=True)
    parser.add_argument("--output_path", type=str, default="output_images", help="Path to save the output images")
    parser.add_argument("--output_format", type=str, choices=["png", "jpg", "jpeg"],
                        help="Output image format (png, jpg, jpeg)")
    parser.add_argument("--FOV", type=int, default=90, help="Field of View in degrees")
    parser.add_argument("--output_width", type=int, default=1000, help="Width of the output image in pixels")
    parser.add_argument("--output_height", type=int, default=1500, help="Height of the output image in pixels")
    parser.add_argument("--pitch", type=check_pitch, default=90, help="Pitch angle in degrees (1-179)")
    parser.add_argument("--yaw_angles", nargs="+", type=int, default=[0, 60, 120, 180, 240, 300],
                        help="List of yaw angles in degrees (0-360). Example: --yaw_angles 0 60 120 180 240 300")
    parser.add_argument("--num_workers", type=int, default=None,
                        help="Number of worker threads. Defaults to 90%% of CPU cores if not specified.")
    return parser


def parse_arguments(argv=None) -> argparse.Namespace:
    args = build_arg_parser().parse_args(argv)
    args.yaw_angles = check_yaw(args.yaw_angles)  # L:301
    return args


def main(argv=None):
    """L:306-388."""
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)s] %(message)s",
                        handlers=[logging.StreamHandler()])
    from tqdm import tqdm

    args = parse_arguments(argv)
    input_path = Path(args.input_path)
    output_path = Path(args.output_path)
    if not input_path.is_dir():
        logging.error(f"Input path {input_path} is not a directory or does not exist.")
        return
    if output_path.exists():
        logging.info(f"Output directory {output_path} already exists.")
    else:
        output_path.mkdir(parents=True, exist_ok=True)
        logging.info(f"Created output directory {output_path}.")

    image_extensions = ["*.jpg", "*.jpeg", "*.png"]
    image_paths = []
    for ext in image_extensions:
        image_paths.extend(input_path.glob(ext))
    if not image_paths:
        logging.warning(f"No images found in {input_path} with extensions {image_extensions}.")
        return

    max_workers = args.num_workers if args.num_workers is not None else max(1, int(os.cpu_count() * 0.9))
    logging.info(f"Using {max_workers} worker(s) for processing.")

    # maps for every yaw, from the first image's size (L:341-363)
    precomputed_mappings = {}
    FOV_rad = np.radians(args.FOV)
    pitch_rad = np.radians(args.pitch)
    sample_pano = _imread_rgb(image_paths[0])
    if sample_pano is None:
        logging.error(f"Failed to read sample image {image_paths[0]} for precomputing mappings.")
        return
    pano_height, pano_width, _ = sample_pano.shape
    for yaw in args.yaw_angles:
        precomputed_mappings[yaw] = precompute_mapping(
            W=args.output_width, H=args.output_height, FOV_rad=FOV_rad, yaw_radian=np.radians(yaw),
            pitch_radian=pitch_rad, pano_width=pano_width, pano_height=pano_height,
        )

    logging.info(f"Starting processing of {len(image_paths)} images with {len(args.yaw_angles)} yaw angles each.")
    with ThreadPoolExecutor(max_workers=max_workers) as executor:
        futures = [executor.submit(process_image_batch, p, args, output_path, precomputed_mappings) for p in image_paths]
        for _ in tqdm(as_completed(futures), total=len(futures), desc="Processing images", unit="image"):
            pass
    logging.info("Processing completed.")


if __name__ == "__main__":
    main()
