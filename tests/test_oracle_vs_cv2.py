"""The pin for the gather oracle, wherever OpenCV is importable: oracle/cv_remap_oracle.c against the real
cv2.remap (the reference pins opencv-python==4.10.0.84, /root/reference/pyproject.toml:11; its calls are
P:192-199, P:212-218 and L:179).  cv2 cannot be installed in the build container or on the GPU box (no wheel,
no network), so there this file is skipped and the gather oracle stays PARITY UNPINNED; on any machine with
opencv-python it runs as is:

    pip install opencv-python-headless==4.10.0.84 && python -m pytest tests/test_oracle_vs_cv2.py -q
"""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="PARITY UNPINNED (gather): cv2 is not importable here; this test pins "
                                        "oracle/cv_remap_oracle.c to the real cv2.remap wherever it is")

from oracle import cpu_ref, maps  # noqa: E402

INTERS = {cpu_ref.INTER_NEAREST: "INTER_NEAREST", cpu_ref.INTER_LINEAR: "INTER_LINEAR", cpu_ref.INTER_CUBIC: "INTER_CUBIC"}
BORDERS = {cpu_ref.BORDER_CONSTANT: "BORDER_CONSTANT", cpu_ref.BORDER_REPLICATE: "BORDER_REPLICATE",
           cpu_ref.BORDER_REFLECT: "BORDER_REFLECT", cpu_ref.BORDER_WRAP: "BORDER_WRAP",
           cpu_ref.BORDER_REFLECT_101: "BORDER_REFLECT_101"}


def _image(rng, h, w, cn):
    a = rng.integers(0, 256, size=(h, w, cn), dtype=np.uint8)
    return a[:, :, 0] if cn == 1 else a


def _maps(rng, sw, sh, ow, oh):
    """Random coordinates plus every class of special value the reference's maps and cv::remap's quantiser meet."""
    U = rng.uniform(-3, sw + 2, size=(oh, ow)).astype(np.float32)
    V = rng.uniform(-3, sh + 2, size=(oh, ow)).astype(np.float32)
    k = np.arange(ow)
    # exact rounding ties of U * 32 (k + 1/64), both parities: cvRound is round-half-even
    U[0, :] = (k % sw) + np.float32(1.0 / 64.0)
    V[0, :] = 1.0
    U[1, :] = (k % sw) + np.float32(3.0 / 64.0)
    V[1, :] = (k % sh) + np.float32(1.0 / 64.0)
    # the image edges: -1, -1/32, 0, w-1, w-1+1/32, w
    edge = np.array([-1.0, -1 / 32, 0.0, sw - 1, sw - 1 + 1 / 32, sw, -1.03125, sw + 0.5], np.float32)
    U[2, :8], V[2, :8] = edge, 0.5
    U[3, :8], V[3, :8] = 0.5, np.array([-1.0, -1 / 32, 0.0, sh - 1, sh - 1 + 1 / 32, sh, -1.03125, sh + 0.5], np.float32)
    # NaN (arccos of 1 + eps at P:162), infinities, values beyond int32 / int16 after the * 32
    U[4, :8] = [np.nan, 1.0, np.inf, -np.inf, 1e9, -1e9, 1100.0, -1100.0]
    V[4, :8] = [1.0, np.nan, 1.0, 1.0, 1.0, 1.0, 2.0, 2.0]
    U[5, :4], V[5, :4] = 1.0, [1e9, -1e9, 40000.0, -40000.0]
    return U, V


@pytest.mark.parametrize("cn", [1, 3, 4])
@pytest.mark.parametrize("border", sorted(BORDERS))
@pytest.mark.parametrize("inter", sorted(INTERS))
def test_remap_equals_cv2(inter, border, cn):
    rng = np.random.default_rng(100 * inter + 10 * border + cn)
    for (sw, sh, ow, oh) in ((37, 23, 64, 16), (256, 128, 80, 48), (5, 4, 32, 8)):
        src = _image(rng, sh, sw, cn)
        U, V = _maps(rng, sw, sh, ow, oh)
        bv = tuple(int(v) for v in rng.integers(1, 255, size=4))
        for border_value in ((0, 0, 0, 0), bv) if border == cpu_ref.BORDER_CONSTANT else ((0, 0, 0, 0),):
            try:
                want = cv2.remap(src, U, V, getattr(cv2, INTERS[inter]), borderMode=getattr(cv2, BORDERS[border]),
                                 borderValue=border_value)
            except cv2.error as e:  # a combination this OpenCV build refuses is not part of the contract
                pytest.skip("cv2 refuses %s / %s: %s" % (INTERS[inter], BORDERS[border], e))
            got = cpu_ref.remap(src, U, V, border, border_value=border_value[:cn], interpolation=inter)
            bad = np.argwhere(got != want)
            assert bad.size == 0, (INTERS[inter], BORDERS[border], cn, (sw, sh), len(bad), bad[:4].tolist())


@pytest.mark.parametrize("pitch", [1, 5, 30, 90, 150, 179])
def test_two_stage_views_equal_two_chained_cv2_remaps(synth, pitch):
    """P:181-221 on the oracle's (reference-pinned) maps: cv2.remap twice == oracle.process_yaw_and_pitchs."""
    pw, ph, ow, oh, fov = 512, 256, 96, 64, 90
    pano = synth.synth_pano(pw, ph, 1700 + pitch, "N")
    for yaw in (0, 30, 77, 359):
        Uy, Vy = maps.yaw_map(pw, ph, yaw)
        rot = cv2.remap(pano, Uy, Vy, cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT, borderValue=(0, 0, 0))
        U, V = maps.pitch_map_deg(ow, oh, pitch, pw, ph, fov)
        want = cv2.remap(rot, U, V, cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT, borderValue=(0, 0, 0))
        got = cpu_ref.process_yaw_and_pitchs(pano, yaw, [pitch], ow, oh, fov)[0]
        assert np.array_equal(cpu_ref.yaw_stage(pano, yaw), rot), (yaw, "stage 1")
        assert np.array_equal(got, want), (yaw, pitch)


def test_legacy_panorama_to_plane_equals_cv2(synth):
    pano = synth.synth_pano(256, 128, 1750, "N")
    U, V = maps.pitch_map_deg(64, 48, 70, 256, 128, 90)
    want = cv2.remap(pano, U, V, cv2.INTER_LINEAR, borderMode=cv2.BORDER_REFLECT)  # L:179
    assert np.array_equal(cpu_ref.panorama_to_plane(pano, U, V), want)
