"""Host-side pitch maps for the tool's --exact mode: COORDINATES only, never pixels.

The reference builds its pitch maps with NumPy on the host and keeps them for the life of the process
(/root/reference/app/panorama_to_plane-pitch.py:114-175, cached by get_pitch_mapping, P:55-73).  Their last
bits come from the host's libm (arccos, arctan2) and BLAS (the 3x3 @ 3xN float32 product, P:155), which no device
evaluation reproduces bit for bit -- and cv2.remap quantises the maps to 1/32 px, so a last-bit difference
moves 0.001-0.017 % of the taps by 1/32 px.  In --exact mode the maps are therefore
evaluated here, with the reference's float32 dtype flow and its one sgemm call in the reference's shape, handed to
the device once per geometry (p2p_job_set_maps / p2p_remap_views_pitch_maps_f64) and every pixel is drawn from
them by the HIP kernels: the views are then the reference's bytes on the same host, also on noise panoramas.

What differs from the reference's own code is only how the arrays are laid out on the way: column and row
vectors are broadcast instead of a meshgrid, and every elementwise step works in place (four W x H float32
temporaries instead of fifteen) -- elementwise float32 operations do not depend on either.

The yaw maps need none of this: P:79-108 uses IEEE operations only and the device's yaw tables are bit-exact.
"""
import threading

import numpy as np

_TWO_PI = 2 * np.pi  # a Python float: "weak" in NumPy's promotion, float32 arrays stay float32 (as in P:164-169)

# the reference's cache and key (P:18, P:62): (output_width, output_height, pitch_angle, pano_width, pano_height, fov_deg)
exact_pitch_mapping_cache = {}
# what the device keeps per pitch LIST: (ow, oh, pitches, pw, ph, fov) -> (U [n_pitch][oh][ow], V, maps_key)
_stacks = {}
_lock = threading.Lock()
_next_key = [1]


def pitch_mapping(W, H, FOV_rad, pitch_radian, pano_width, pano_height):
    """(U, V) float32 (H, W) with the values of precompute_pitch_mapping (P:114-175)."""
    c, s = np.cos(pitch_radian), np.sin(pitch_radian)
    R = np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float32)  # P:142-149
    return _rotated_ray_map(W, H, FOV_rad, R, pano_width, pano_height)


def legacy_mapping(W, H, FOV_rad, yaw_radian, pitch_radian, pano_width, pano_height):
    """(U, V) float32 (H, W) with the values of the LEGACY tool's precompute_mapping
    (/root/reference/app/legacy/panorama_to_plane.py:47-157): the same ray map under the combined rotation
    R_pitch @ R_yaw of get_rotation_matrix (L:21-45), two float32 3x3 arrays multiplied with np.dot."""
    cy, sy = np.cos(yaw_radian), np.sin(yaw_radian)
    cp, sp = np.cos(pitch_radian), np.sin(pitch_radian)
    R_yaw = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float32)    # L:32-36
    R_pitch = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], dtype=np.float32)  # L:38-42
    return _rotated_ray_map(W, H, FOV_rad, np.dot(R_pitch, R_yaw), pano_width, pano_height)  # L:45


def _rotated_ray_map(W, H, FOV_rad, R, pano_width, pano_height):
    """What both tools' builders share (P:119-175 / L:95-157): pinhole rays, normalised, rotated by the float32 3x3 R,
    mapped to panorama coordinates."""
    W, H = int(W), int(H)
    focal = (0.5 * W) / np.tan(FOV_rad / 2)                 # P:119, float64 scalar
    x = np.arange(W, dtype=np.float32) - (W / 2.0)          # P:129 for one row of pixels
    y = (H / 2.0) - np.arange(H, dtype=np.float32)          # P:130 for one column
    z = np.float32(focal)                                   # P:131 (full_like casts the scalar to float32)
    norm = (x * x)[None, :] + (y * y)[:, None]              # P:134: (x**2 + y**2) + z**2, then the root
    norm += z * z
    np.sqrt(norm, out=norm)
    rays = np.empty((3, H * W), dtype=np.float32)           # P:152: (x, y, z) / norm stacked as 3 x (W*H)
    np.divide(x[None, :], norm, out=rays[0].reshape(H, W))  # P:137-139
    np.divide(y[:, None], norm, out=rays[1].reshape(H, W))
    np.divide(z, norm, out=rays[2].reshape(H, W))
    del norm
    rot = R @ rays                                          # P:155 / L:123: ONE float32 gemm of the reference's shape
    del rays
    x_rot, y_rot, z_rot = rot.reshape(3, H, W)              # P:158
    with np.errstate(invalid="ignore"):                     # (z_rot may round above 1: NaN, a black pixel, P:162)
        V = np.arccos(z_rot)                                # P:162 theta'
    U = np.arctan2(y_rot, x_rot)                            # P:164 phi'
    del rot, x_rot, y_rot, z_rot
    np.remainder(U, _TWO_PI, out=U)
    U *= pano_width                                         # P:167
    U /= _TWO_PI
    V *= pano_height                                        # P:169
    V /= np.pi
    np.clip(U, 0, pano_width - 1, out=U)                    # P:172-173 (NaN stays NaN)
    np.clip(V, 0, pano_height - 1, out=V)
    return U, V


def get_pitch_mapping(output_width, output_height, pitch_angle, pano_width, pano_height, fov_deg=90):
    """get_pitch_mapping of P:55-73 over the builder above, same key, same np.radians of the degree arguments."""
    key = (output_width, output_height, pitch_angle, pano_width, pano_height, fov_deg)
    with _lock:
        hit = exact_pitch_mapping_cache.get(key)
    if hit is None:
        hit = pitch_mapping(output_width, output_height, np.radians(fov_deg), np.radians(pitch_angle), pano_width, pano_height)
        with _lock:
            hit = exact_pitch_mapping_cache.setdefault(key, hit)
    return hit


def pitch_map_stack(output_width, output_height, pitch_angles, pano_width, pano_height, fov_deg=90):
    """The maps of a whole pitch list as the device takes them: (U, V, maps_key), U / V float32 [n_pitch][H][W],
    maps_key a process-unique non-zero integer that names exactly these arrays -- a job that holds them under that key
    need not be sent them again."""
    pitches = tuple(pitch_angles)
    key = (output_width, output_height, pitches, pano_width, pano_height, fov_deg)
    with _lock:
        hit = _stacks.get(key)
    if hit is not None:
        return hit
    maps = [get_pitch_mapping(output_width, output_height, p, pano_width, pano_height, fov_deg) for p in pitches]
    U = np.ascontiguousarray(np.stack([m[0] for m in maps])) if maps else np.empty((0, int(output_height), int(output_width)), np.float32)
    V = np.ascontiguousarray(np.stack([m[1] for m in maps])) if maps else U.copy()
    with _lock:
        hit = _stacks.get(key)
        if hit is None:
            hit = _stacks[key] = (U, V, _next_key[0])
            _next_key[0] += 1
    return hit


def clear():
    with _lock:
        exact_pitch_mapping_cache.clear()
        _stacks.clear()
