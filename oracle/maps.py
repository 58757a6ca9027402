"""oracle/maps.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy restatement of the reference's two coordinate-map builders, with the same
dtype flow (NumPy >= 2 / NEP 50 promotion rules), so that the result is the
reference's float32 map bit for bit on the same host:

  yaw map   : /root/reference/app/panorama_to_plane-pitch.py:79-108
  pitch map : /root/reference/app/panorama_to_plane-pitch.py:114-175
  wrappers  : /root/reference/app/panorama_to_plane-pitch.py:42-73 (radians conversion)

PINNED: tests/golden/maps_*.npz were produced by importing the reference's own
functions in the build container (tests/golden/make_golden_maps.py) and
tests/test_oracle_maps.py checks this restatement against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product never does.
"""
import numpy as np

TWO_PI = 2 * np.pi  # python float, "weak" under NEP 50 (P:95, P:98, P:101, P:164, P:167)


def yaw_column_table(pano_width, yaw_angle):
    """U_yaw[0, :] of P:79-108: one float32 row of length pano_width.

    Every row of the reference's U is this row and V[y, x] == y (P:102), because
    the formula depends on the column index only.
    dtype flow: phi float32 (P:95) -> + np.float64 yaw (P:85, P:98) promotes to
    float64 -> % 2pi, * pw / 2pi in float64 (P:98-101) -> clip -> float32 (P:105).
    """
    yaw_radians = np.radians(yaw_angle)  # np.float64 scalar (P:85)
    u = np.arange(pano_width, dtype=np.float32)
    phi = (TWO_PI * u / pano_width).astype(np.float32)  # P:95
    phi_rotated = (phi + yaw_radians) % TWO_PI  # float64, P:98
    U = (phi_rotated * pano_width) / TWO_PI  # P:101
    return np.clip(U, 0, pano_width - 1).astype(np.float32)  # P:105


def yaw_map(pano_width, pano_height, yaw_angle):
    """Full (U, V) pair of P:79-108, each (pano_height, pano_width) float32."""
    row = yaw_column_table(pano_width, yaw_angle)
    U = np.broadcast_to(row, (pano_height, pano_width)).copy()
    V = np.broadcast_to(
        np.arange(pano_height, dtype=np.float32)[:, None], (pano_height, pano_width)
    ).copy()
    return U, V


def pitch_map(W, H, FOV_rad, pitch_radian, pano_width, pano_height, clip_u=True):
    """(U, V) of P:114-175, each (H, W) float32.  Arguments as P:114.
    clip_u=False is NOT the reference: it leaves the azimuth unclipped in [0, pano_width), which is what the
    build's opt-in float pixel path (true wrap-around at the seam) resamples at; tests of that path use it."""
    focal = (0.5 * W) / np.tan(FOV_rad / 2)  # P:119 (float64 scalar)
    u, v = np.meshgrid(
        np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy"
    )  # P:122-126
    x = u - (W / 2.0)  # P:129
    y = (H / 2.0) - v  # P:130
    z = np.full_like(x, focal, dtype=np.float32)  # P:131
    norm = np.sqrt(x**2 + y**2 + z**2)  # P:134
    vec = np.stack((x / norm, y / norm, z / norm), axis=0).reshape(3, -1)  # P:137-152
    c, s = np.cos(pitch_radian), np.sin(pitch_radian)
    R = np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float32)  # P:142-149
    x_rot, y_rot, z_rot = (R @ vec).reshape(3, H, W)  # P:155-158 (float32 sgemm)
    with np.errstate(invalid="ignore"):
        theta = np.arccos(z_rot).astype(np.float32)  # P:162 (NaN when z_rot rounds above 1)
    phi = (np.arctan2(y_rot, x_rot) % TWO_PI).astype(np.float32)  # P:164
    U = (phi * pano_width) / TWO_PI  # P:167
    V = (theta * pano_height) / np.pi  # P:169
    if clip_u:
        U = np.clip(U, 0, pano_width - 1).astype(np.float32)  # P:172
    else:
        U = np.where(U >= pano_width, U - pano_width, U).astype(np.float32)
    V = np.clip(V, 0, pano_height - 1).astype(np.float32)  # P:173 (NaN stays NaN)
    return U, V


def pitch_map_deg(output_width, output_height, pitch_angle, pano_width, pano_height, fov_deg=90, clip_u=True):
    """get_pitch_mapping()'s argument convention (P:55-73): degrees in, np.radians() applied."""
    return pitch_map(
        output_width,
        output_height,
        np.radians(fov_deg),
        np.radians(pitch_angle),
        pano_width,
        pano_height,
        clip_u=clip_u,
    )


# ------------------------------------------------------------------------------------------------
# legacy tool: one combined yaw + pitch rotation, one remap
#   /root/reference/app/legacy/panorama_to_plane.py ("L"): get_rotation_matrix L:21-45,
#   precompute_mapping L:47-157.  Pinned by tests/golden/legacy_maps_golden.npz
#   (tests/golden/make_golden_legacy.py imports the reference's functions).
# ------------------------------------------------------------------------------------------------
def legacy_rotation_matrix(yaw_radian, pitch_radian):
    """L:21-45: float32 R_pitch @ R_yaw (np.dot of two float32 3x3 arrays)."""
    cy, sy = np.cos(yaw_radian), np.sin(yaw_radian)
    cp, sp = np.cos(pitch_radian), np.sin(pitch_radian)
    R_yaw = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float32)  # L:32-36
    R_pitch = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], dtype=np.float32)  # L:38-42
    return np.dot(R_pitch, R_yaw)  # L:45


def legacy_map(W, H, FOV_rad, yaw_radian, pitch_radian, pano_width, pano_height):
    """(U, V) of L:47-157, each (H, W) float32.  Arguments as L:48."""
    focal = (0.5 * W) / np.tan(FOV_rad / 2)  # L:95
    u, v = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")  # L:98
    u = u.astype(np.float32)  # L:101-102
    v = v.astype(np.float32)
    x = u - (W / 2.0)  # L:107
    y = (H / 2.0) - v  # L:108
    z = np.full_like(x, focal, dtype=np.float32)  # L:109
    norm = np.sqrt(x**2 + y**2 + z**2)  # L:113
    vec = np.stack((x / norm, y / norm, z / norm), axis=0)  # L:114-116, L:122
    R = legacy_rotation_matrix(yaw_radian, pitch_radian)  # L:121
    x_rot, y_rot, z_rot = (R @ vec.reshape(3, -1)).reshape(3, H, W)  # L:123-125
    with np.errstate(invalid="ignore"):
        theta = np.arccos(z_rot).astype(np.float32)  # L:130
    phi = (np.arctan2(y_rot, x_rot) % TWO_PI).astype(np.float32)  # L:145
    U = (phi * pano_width) / TWO_PI  # L:157-158
    V = (theta * pano_height) / np.pi
    U = np.clip(U, 0, pano_width - 1).astype(np.float32)  # L:161-162
    V = np.clip(V, 0, pano_height - 1).astype(np.float32)
    return U, V
