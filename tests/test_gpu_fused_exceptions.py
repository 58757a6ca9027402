"""The fused mode's KNOWN exceptions to "+-1 per channel", pinned (arithmetic contract).  With device-evaluated
maps (the default mode) three things are allowed to differ from the oracle's NumPy maps by more than one level, and
nothing else is:
  * the pole pixel -- one output pixel per view whose ray has x = 0 exactly and points less than a source row past a
    pole: its azimuth is arctan2(+-0, z'), z' a rounding residue of the pitch rotation (P:155's sgemm, P:164), and the
    SIGN of that residue picks one of two columns half a panorama apart.  Any colour of the pole row may come out;
  * the seam row (round 6) -- an output ROW whose rays, rotated by the pitch, have y' = 0 exactly (tan(pitch) * focal length is
    that row's y: FOV 120, pitch 30, W / 2 divisible by 3), so that the azimuth of its pixels with x' > 0 is arctan2(+-0, x')
    and the sign of the residue puts a pixel at U = 0 or at U = 2 pi, clipped to the last column (P:164-173; the reference
    does not interpolate across the seam): column 0 or column pw - 1 of the panorama, neighbours on the sphere, a few levels
    apart.  The reference's own choice per pixel hangs on its BLAS;
  * steep gradients under a wide FOV on a small panorama: a coordinate that lands on the other side of a 1/32-pixel
    rounding tie moves a byte by 2-3 levels where neighbouring source pixels differ by a dozen -- on at most 1e-5 of
    the bytes of a view set.
A regression that widens either set fails here.  The cases are the three that round 4's fuzz_fused.py run met
(profiles/r04_fuzz_round_final_tree.txt: cases 125, 51 and 39 of seed 74; panorama seed = 700 + case)."""
import numpy as np
import pytest

from _util import oracle_maps, oracle_views

pytestmark = pytest.mark.gpu

CASES = [
    # name, pw, ow, oh, fov, yaws, pitches, panorama seed
    ("pole pixel", 1024, 372, 183, 90, [285, 252], [9, 54], 825),
    ("fov 150 at the nadir", 1024, 188, 230, 150, [358, 3, 187], [105, 179], 751),
    ("fov 20 next to both poles", 512, 135, 353, 20, [116, 35, 182], [10, 157], 739),
    # round 6's last fuzz round (profiles/r06_fuzz_round.txt: case 35 of seed 313): row 93 of the pitch-30 view is the seam
    ("the seam row", 1024, 474, 344, 120, [297, 314], [30, 129], 735),
]


def pole_pixels(U, V, ow, ph):
    """Mask of the output pixels that are 'the pole pixel' of a pitch view: x = u - W/2 == 0 (P:129) and the oracle's
    own V within one source row of a pole (P:169-173)."""
    m = np.zeros(V.shape, bool)
    if ow % 2 == 0:
        col = ow // 2
        v = V[:, col]
        with np.errstate(invalid="ignore"):
            m[:, col] = (v < 1.0) | (v > ph - 2.0) | np.isnan(v)
    return m


def seam_pixels(U, pw):
    """Mask of the output pixels the oracle's own map puts ON the seam: U = 0 or U = pw - 1 (clipped), P:164-173."""
    with np.errstate(invalid="ignore"):
        return (U < 1.0 / 32) | (U > pw - 1 - 1.0 / 32)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fused_mode_differs_by_more_than_one_only_where_it_is_known_to(gpu, synth, case):
    _, pw, ow, oh, fov, yaws, pitches, seed = case
    ph = pw // 2
    pano = synth.synth_pano(pw, ph, seed, "S")
    got = gpu.remap_views(pano, yaws, pitches, fov, ow, oh)
    want = oracle_views(pano, yaws, pitches, ow, oh, fov)
    _, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16)).max(axis=-1)  # [yaw][pitch][oh][ow]
    outside = 0
    for pi in range(len(pitches)):
        pole = pole_pixels(U[pi], V[pi], ow, ph)
        seam = seam_pixels(U[pi], pw)
        # (the seam is a property of a whole row, or of nothing: a view whose map merely touches column 0 is not excused)
        seam_row = seam & (seam.sum(axis=1, keepdims=True) >= ow // 4)
        for yi in range(len(yaws)):
            big = d[yi, pi] > 1
            at_pole = big & pole
            assert at_pole.sum() <= 1, ("more than one pole pixel differs in a view", yaws[yi], pitches[pi], np.argwhere(at_pole)[:4].tolist())
            at_seam = big & seam_row & ~pole
            assert d[yi, pi][seam_row].max(initial=0) <= 8, ("the seam row: more than two neighbouring columns apart", yaws[yi], pitches[pi])
            rest = big & ~pole & ~seam_row
            outside += int(rest.sum())
            assert d[yi, pi][~pole & ~seam_row].max() <= 4, ("a difference beyond a flipped 1/32-pixel coordinate", yaws[yi], pitches[pi],
                                                              int(d[yi, pi][~pole & ~seam_row].max()), np.argwhere(rest)[:4].tolist())
            if case[0] == "the seam row" and pitches[pi] == 30:
                assert at_seam.sum() >= 1 and np.unique(np.argwhere(at_seam)[:, 0]).size == 1  # (not vacuous: one row, and it differs)
    n_px = d.size
    assert outside <= max(1, int(1e-5 * n_px * 3)), ("fused mode: pixels off by more than 1 outside the pole pixel", outside, n_px)


def test_the_pole_pixel_case_has_its_pole_pixel(gpu, synth):
    """The mask above is not vacuous: the round-4 case does contain a pixel with x = 0 within a row of the pole, in the
    pitch-9 view, and the caller-map path (the oracle's own maps) draws even that pixel byte for byte."""
    _, pw, ow, oh, fov, yaws, pitches, seed = CASES[0]
    ph = pw // 2
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    # (pitch 9: the column x = W / 2 passes within a source row of the pole on three output rows -- V = 0.82, 0.06, 0.90 --
    # and the middle one is the pixel whose azimuth hangs on the sign of a residue; pitch 54 is nowhere near a pole)
    assert 1 <= pole_pixels(U[0], V[0], ow, ph).sum() <= 3 and pole_pixels(U[1], V[1], ow, ph).sum() == 0
    pano = synth.synth_pano(pw, ph, seed, "S")
    exact = gpu.remap_views_maps(pano, rows, U, V)
    assert np.array_equal(exact, oracle_views(pano, yaws, pitches, ow, oh, fov))
