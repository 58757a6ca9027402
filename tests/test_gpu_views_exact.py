"""GPU parity, integer path: with the oracle's own float maps handed to the kernel
(p2p_remap_views_maps_u8 / p2p_job_set_maps) every output byte must equal the CPU restatement of
the reference's two cv2.remap stages (P:181-221).  Bit-exact, no tolerance."""
import numpy as np
import pytest

from _util import oracle_maps, oracle_views

pytestmark = pytest.mark.gpu


def _check(gpu, pano, yaws, pitches, ow, oh, fov=90):
    ph, pw = pano.shape[:2]
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    got = gpu.remap_views_maps(pano, rows, U, V)
    want = oracle_views(pano, yaws, pitches, ow, oh, fov)
    assert got.shape == want.shape
    bad = np.argwhere(got != want)
    assert bad.size == 0, (len(bad), bad[:5], got[tuple(bad[0])], want[tuple(bad[0])])


def test_cfg1_noise(gpu, synth):
    # BASELINE config 1: 2048x1024 -> 512x512, FOV 90, yaw 0, pitch 90
    _check(gpu, synth.synth_pano(2048, 1024, 1000, "N"), [0], [90], 512, 512)


def test_cfg1_bandlimited(gpu, synth):
    _check(gpu, synth.synth_pano(2048, 1024, 1000, "S"), [0], [90], 512, 512)


def test_fractional_yaw_and_pitch(gpu, synth):
    # yaw 30 (non-integer column shift), pitch 60
    _check(gpu, synth.synth_pano(2048, 1024, 1001, "N"), [30], [60], 512, 512)


def test_many_yaws_pitches_order(gpu, synth):
    _check(gpu, synth.synth_pano(1024, 512, 1002, "N"), [0, 1, 45, 77, 90, 359, 360, -30, 400],
           [30, 60, 90, 120, 150], 200, 120)


def test_pole_views_use_direct_gather(gpu, synth):
    # pitch 1 / 179 look at the poles: the footprint spans every column (seam, no wrap) -> global gather path
    _check(gpu, synth.synth_pano(1024, 512, 1003, "N"), [0, 77], [1, 5, 175, 179], 160, 160)


def test_nan_pixel_is_black(gpu, synth):
    # pitch 5 at 1920x1080 has one NaN coordinate (arccos of 1+eps, P:162) -> borderValue 0
    pano = synth.synth_pano(2048, 1024, 1004, "N")
    pano[pano == 0] = 1  # so that a black output pixel can only be the NaN one
    ph, pw = pano.shape[:2]
    rows, U, V = oracle_maps([0], [5], 1920, 1080, pw, ph, 90)
    got = gpu.remap_views_maps(pano, rows, U, V)
    want = oracle_views(pano, [0], [5], 1920, 1080, 90)
    assert np.array_equal(got, want)
    nan_px = np.argwhere(np.isnan(V[0]))
    for (r, c) in nan_px:
        assert (got[0, 0, r, c] == 0).all()


@pytest.mark.parametrize("ow,oh", [(33, 7), (1, 1), (5, 300), (130, 9), (257, 64)])
def test_odd_output_sizes_byte_store_path(gpu, synth, ow, oh):
    _check(gpu, synth.synth_pano(512, 256, 1005, "N"), [0, 123], [45, 90], ow, oh)


@pytest.mark.parametrize("pw,ph", [(64, 32), (100, 50), (333, 111), (17, 9)])
def test_odd_panorama_sizes(gpu, synth, pw, ph):
    _check(gpu, synth.synth_pano(pw, ph, 1006, "N"), [0, 10, 200], [30, 90, 150], 64, 48, fov=120)


def test_wide_fov_and_narrow_fov(gpu, synth):
    pano = synth.synth_pano(1024, 512, 1007, "N")
    _check(gpu, pano, [15], [90, 60], 128, 128, fov=150)
    _check(gpu, pano, [15], [90, 60], 128, 128, fov=20)


def test_strided_panorama_rows(gpu, synth):
    big = synth.synth_pano(600, 200, 1008, "N")
    view = big[:, 44:556]  # row stride 1800 bytes, width 512
    assert not view.flags.c_contiguous
    _check(gpu, view, [0, 50], [90], 96, 64)


def test_resident_job_multi_pano(gpu, synth):
    pw, ph, ow, oh = 512, 256, 96, 64
    yaws, pitches = [0, 30, 200], [60, 120]
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, pw, ph, 3, yaws, pitches, 90, ow, oh)
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, 90)
    job.set_maps(rows, U, V)
    panos = [synth.synth_pano(pw, ph, 1010 + i, "N") for i in range(3)]
    for i, p in enumerate(panos):
        job.set_pano(i, p)
    job.run()
    for i, p in enumerate(panos):
        assert np.array_equal(job.get_views(i), oracle_views(p, yaws, pitches, ow, oh, 90))
    with pytest.raises(gpu.P2PError):
        job.kernel_ms()          # launch timing is off unless asked for (no timing events, none recorded)
    job.time_launches(2)
    job.run()
    assert job.kernel_ms() > 0 and job.info()["timing_events"] == 4
    job.close()
    ctx.close()


def test_flickering_yaw_fraction_per_column_weights(gpu, synth):
    # yaw 14 on an 8192-wide panorama: 32 * 14 * 8192 / 360 sits within float32 noise of a rounding
    # tie, so the 1/32-px fraction differs from column to column (YawDesc mode 1)
    from oracle import cpu_ref, maps
    row = maps.yaw_column_table(8192, 14)
    _, _, fxq, _ = cpu_ref.quantise_maps(row[None], np.zeros((1, 8192), np.float32))
    assert len(np.unique(fxq)) >= 2
    _check(gpu, synth.synth_pano(8192, 512, 1011, "N"), [14, 59, 0], [80, 100], 256, 144)


def test_caller_yaw_rows_that_are_not_a_shift(gpu, synth):
    # p2p_remap_views_maps_u8 accepts any in-range yaw row; a mirrored / stretched row is not a circular
    # shift (YawDesc mode 2) and must still equal two chained cv2.remap calls
    from oracle import cpu_ref, maps
    pw, ph, ow, oh = 512, 256, 96, 64
    pano = synth.synth_pano(pw, ph, 1012, "N")
    x = np.arange(pw, dtype=np.float32)
    rows = np.stack([pw - 1 - x, np.clip(x * 0.731 + 3.3, 0, pw - 1), maps.yaw_column_table(pw, 40)]).astype(np.float32)
    U, V = maps.pitch_map_deg(ow, oh, 75, pw, ph, 90)
    got = gpu.remap_views_maps(pano, rows, U[None], V[None])
    Vy = np.broadcast_to(np.arange(ph, dtype=np.float32)[:, None], (ph, pw))
    for k in range(3):
        rot = cpu_ref.remap(pano, np.broadcast_to(rows[k], (ph, pw)), Vy, cpu_ref.BORDER_CONSTANT)
        want = cpu_ref.remap(rot, U, V, cpu_ref.BORDER_CONSTANT)
        assert np.array_equal(got[k, 0], want), k


def test_general_caller_pitch_maps_with_border_taps(gpu, synth):
    # maps that leave the panorama (not produced by the reference, but legal cv2.remap input):
    # BORDER_CONSTANT 0 taps, partially and fully outside pixels
    from oracle import cpu_ref, maps
    pw, ph, ow, oh = 256, 128, 80, 48
    pano = synth.synth_pano(pw, ph, 1013, "N")
    rng = np.random.default_rng(5)
    U = rng.uniform(-3, pw + 2, size=(1, oh, ow)).astype(np.float32)
    V = rng.uniform(-3, ph + 2, size=(1, oh, ow)).astype(np.float32)
    U[0, :4, :8] = [[-1.0, -0.5, pw - 1, pw - 0.5, pw, 0, 0.25, -1.03125]] * 4
    V[0, :4, :8] = np.array([[-1.0], [ph - 1], [ph - 0.5], [0.0]], np.float32)
    rows = maps.yaw_column_table(pw, 77)[None]
    got = gpu.remap_views_maps(pano, rows, U, V)
    rot = cpu_ref.yaw_stage(pano, 77)
    want = cpu_ref.remap(rot, U[0], V[0], cpu_ref.BORDER_CONSTANT)
    assert np.array_equal(got[0, 0], want)


def test_pair_chunking_across_workgroups(gpu, synth, monkeypatch, p2p_env):
    # P2P_PAIRS_PER_BLOCK forces the (panorama, yaw) pairs of a tile to be split over several workgroups
    # (grid.z chunks, as happens by itself for small outputs and long yaw sweeps); results must not depend on it
    pw, ph, ow, oh = 512, 256, 96, 64
    yaws, pitches = list(range(0, 360, 24)), [70, 110]  # 15 yaws
    panos = [synth.synth_pano(pw, ph, 1020 + i, "N") for i in range(2)]
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, 90)
    want = [oracle_views(p, yaws, pitches, ow, oh, 90) for p in panos]
    for ppb in ("1", "4", "7", "64"):
        p2p_env("P2P_PAIRS_PER_BLOCK", ppb)
        ctx = gpu.Context(0)
        job = gpu.Job(ctx, pw, ph, 2, yaws, pitches, 90, ow, oh)
        job.set_maps(rows, U, V)
        for i, p in enumerate(panos):
            job.set_pano(i, p)
        job.run()
        for i in range(2):
            assert np.array_equal(job.get_views(i), want[i]), (ppb, i)
        job.close()
        ctx.close()


def test_long_yaw_sweep_config5_style(gpu, synth):
    # config 5 geometry at reduced size: a 1-degree yaw sweep (whole-column, fractional and flickering shifts mixed)
    pano = synth.synth_pano(2048, 1024, 1030, "N")
    _check(gpu, pano, list(range(0, 360, 7)) + [359, 360, 361], [90], 240, 136)


def test_randomised_configurations(gpu, synth):
    # seeded sweep over panorama / view sizes, FOVs, yaw and pitch lists: every byte equals the oracle
    rng = np.random.default_rng(4242)
    for case in range(24):
        pw = int(rng.choice([64, 128, 200, 256, 372, 512, 1000, 1024]))
        ph = int(rng.choice([32, 64, 100, 128, 256, 500]))
        ow, oh = int(rng.integers(1, 160)), int(rng.integers(1, 100))
        fov = int(rng.integers(10, 170))
        yaws = [int(v) for v in rng.integers(-400, 800, size=int(rng.integers(1, 5)))]
        pitches = [int(v) for v in rng.integers(1, 180, size=int(rng.integers(1, 4)))]
        pano = synth.synth_pano(pw, ph, 5000 + case, "N")
        ph_, pw_ = pano.shape[:2]
        rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw_, ph_, fov)
        got = gpu.remap_views_maps(pano, rows, U, V)
        want = oracle_views(pano, yaws, pitches, ow, oh, fov)
        assert np.array_equal(got, want), dict(case=case, pw=pw, ph=ph, ow=ow, oh=oh, fov=fov, yaws=yaws, pitches=pitches)


def test_randomised_large_panoramas_mixed_chunks_and_sub_tiles(gpu, synth):
    # larger sources (footprints that outgrow LDS at low pitch), many yaws per workgroup incl. the odd ones that
    # leave the tight loop, several panoramas per job
    rng = np.random.default_rng(777)
    for case in range(6):
        pw = int(rng.choice([2048, 4096, 8192]))
        ph = pw // 2
        ow, oh = int(rng.integers(200, 700)), int(rng.integers(100, 400))
        fov = int(rng.choice([60, 90, 110]))
        yaws = sorted({int(v) for v in rng.integers(0, 360, size=int(rng.integers(13, 40)))} | {14, 59})
        pitches = [int(v) for v in rng.integers(15, 166, size=2)]
        n_panos = int(rng.integers(1, 3))
        panos = [synth.synth_pano(pw, ph, 6000 + 10 * case + i, "N") for i in range(n_panos)]
        rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
        ctx = gpu.Context(0)
        job = gpu.Job(ctx, pw, ph, n_panos, yaws, pitches, fov, ow, oh)
        for i, p in enumerate(panos):
            job.set_pano(i, p)
        job.set_maps(rows, U, V)
        job.run()
        for i, p in enumerate(panos):
            got = job.get_views(i)
            # the oracle is slow at these sizes: check three of the yaws
            for yi in (0, len(yaws) // 2, len(yaws) - 1):
                want = oracle_views(p, [yaws[yi]], pitches, ow, oh, fov)
                assert np.array_equal(got[yi], want[0]), dict(case=case, pw=pw, ow=ow, oh=oh, fov=fov, yaw=yaws[yi], pitches=pitches)
        job.close()
        ctx.close()


@pytest.mark.parametrize("pitch", [12, 30, 45, 56, 135, 150, 168])
def test_large_footprints_split_into_sub_blocks(gpu, synth, pitch):
    """Towards a pole a 32x16 tile's footprint outgrows the LDS buffers; the kernel then draws the tile as
    32x8 or 16x8 sub-tiles with their own footprints (and gathers directly only where a pole sits inside a
    sub-tile).  4096x2048 -> 960x540 has all of those cases for these pitches; several yaw kinds per launch."""
    pano = synth.synth_pano(4096, 2048, 1200 + pitch, "N")
    _check(gpu, pano, [0, 45, 77, 200], [pitch], 960, 540)


def test_sub_blocks_mixed_with_plain_tiles_in_one_launch(gpu, synth):
    # the reference's default pitch list (P:428) on one panorama, odd output size (byte store path too)
    pano = synth.synth_pano(2048, 1024, 1300, "N")
    _check(gpu, pano, [0, 13, 90], [30, 60, 90, 120, 150], 640, 480)
    _check(gpu, pano, [13], [30, 150], 333, 250)
    _check(gpu, pano, [5, 359], [25], 640, 480, fov=120)


def test_single_yaw_jobs_with_several_panoramas(gpu, synth):
    """n_yaw == 1 makes the pair -> panorama constant ceil(2^32 / n_yaw) overflow 32 bits (found by
    tests/fuzz/fuzz_parity.py: every panorama but the first came out as the first)."""
    panos = [synth.synth_pano(512, 256, 1400 + i, "N") for i in range(4)]
    for yaws in ([14], [0], [200]):
        rows, U, V = oracle_maps(yaws, [60, 107], 169, 81, 512, 256, 90)
        ctx = gpu.Context(0)
        job = gpu.Job(ctx, 512, 256, len(panos), yaws, [60, 107], 90, 169, 81)
        for i, p in enumerate(panos):
            job.set_pano(i, p)
        job.set_maps(rows, U, V)
        job.run()
        for i, p in enumerate(panos):
            assert np.array_equal(job.get_views(i), oracle_views(p, yaws, [60, 107], 169, 81, 90)), (yaws, i)
        job.close()
        ctx.close()


def test_rest_pair_list_runs_over_several_chunks_and_panoramas(gpu, synth):
    # 3 panoramas x the 6 flickering yaws of an 8192-wide panorama (14 + 45 k: per-column weights) = 18 pairs for
    # the rest kernel's own pair list, i.e. two chunks of it (16 pairs per workgroup), next to plain yaws that the
    # main kernel draws
    pw, ph, ow, oh = 8192, 512, 256, 144
    yaws, pitches = [14, 0, 59, 104, 30, 149, 194, 239], [80, 100]
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, pw, ph, 3, yaws, pitches, 90, ow, oh)
    rows, U, V = oracle_maps(yaws, pitches, ow, oh, pw, ph, 90)
    job.set_maps(rows, U, V)
    panos = [synth.synth_pano(pw, ph, 1030 + i, "N") for i in range(3)]
    for i, p in enumerate(panos):
        job.set_pano(i, p)
    job.run()
    for i, p in enumerate(panos):
        assert np.array_equal(job.get_views(i), oracle_views(p, yaws, pitches, ow, oh, 90)), i
    job.close()
    ctx.close()
