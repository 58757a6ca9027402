"""360-to-planer-images_amd -- MI355X (gfx950) view synthesis for 360-degree panoramas.

The directory name is not a Python identifier; import it with

    import importlib; p2p = importlib.import_module("360-to-planer-images_amd")

or run the CLI file directly: python 360-to-planer-images_amd/panorama_to_plane_pitch.py --input_path ...

Contents: csrc/ (HIP kernels + C ABI, built into libp2p_hip.so), _native.py (ctypes binding),
panorama_to_plane_pitch.py (mirror of the reference's current tool), panorama_to_plane.py
(mirror of its legacy tool: combined-rotation maps, one remap, its own CLI).
"""
from . import _native
from .panorama_to_plane import get_rotation_matrix, interpolate_color, panorama_to_plane, precompute_mapping
from .panorama_to_plane_pitch import (
    check_pitch,
    get_pitch_mapping,
    get_version,
    get_yaw_mapping,
    main,
    precompute_pitch_mapping,
    precompute_yaw_mapping,
    process_single_image,
    process_views,
    process_yaw_and_pitchs,
    set_device,
    set_devices,
    set_exact,
    set_quality,
)

__all__ = [
    "check_pitch", "get_pitch_mapping", "get_version", "get_yaw_mapping", "main",
    "precompute_pitch_mapping", "precompute_yaw_mapping", "process_single_image", "process_views",
    "process_yaw_and_pitchs", "set_device", "set_devices", "set_exact", "set_quality", "panorama_to_plane", "interpolate_color",
    "get_rotation_matrix", "precompute_mapping",
]
