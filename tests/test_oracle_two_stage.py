"""The restated two-stage view synthesis (P:181-221): properties the reference's geometry implies."""
import numpy as np

from oracle import cpu_ref, maps


def test_yaw_stage_is_a_per_column_two_tap_filter(synth):
    # P:192-199 with V == y: rot[y][x] = ((32-f) src[y][i] + f src[y][i+1] + 16) >> 5, (i, f) from column x
    pano = synth.synth_pano(256, 16, 1000, "N")
    for yaw in (0, 30, 77, 359, -30):
        rot = cpu_ref.yaw_stage(pano, yaw)
        row = maps.yaw_column_table(256, yaw)
        ix, _, fx, _ = cpu_ref.quantise_maps(row[None], np.zeros((1, 256), np.float32))
        i, f = ix[0].astype(np.int64), fx[0].astype(np.int64)
        assert ((i < 255) | (f == 0)).all()  # P:105's clip: the right tap never leaves the row with weight
        a = pano[:, i].astype(np.int64)
        b = pano[:, np.minimum(i + 1, 255)].astype(np.int64)
        want = ((32 - f)[None, :, None] * a + f[None, :, None] * b + 16) >> 5
        assert np.array_equal(rot, want.astype(np.uint8))


def test_integer_yaw_shift_equals_roll(synth):
    # 45 degrees on a 2048-wide panorama = 256 whole columns: stage 1 is a pure circular shift
    pano = synth.synth_pano(2048, 64, 1001, "N")
    rot = cpu_ref.yaw_stage(pano, 45)
    assert np.array_equal(rot, np.roll(pano, -256, axis=1))
    a = cpu_ref.process_yaw_and_pitchs(pano, 45, [90], 64, 48, 90)[0]
    b = cpu_ref.process_yaw_and_pitchs(np.roll(pano, -256, axis=1), 0, [90], 64, 48, 90)[0]
    assert np.array_equal(a, b)


def test_clamp_column_has_no_wrap_interpolation(synth):
    pano = synth.synth_pano(2048, 8, 1002, "N")
    rot = cpu_ref.yaw_stage(pano, 30)
    assert np.array_equal(rot[:, 1877], pano[:, 2047])  # clamped to pw-1, pure copy
    want = (11 * pano[:, 0].astype(np.int64) + 21 * pano[:, 1].astype(np.int64) + 16) >> 5
    assert np.array_equal(rot[:, 1878], want.astype(np.uint8))


def test_constant_panorama_gives_constant_views(synth):
    pano = np.full((128, 256, 3), (10, 128, 250), np.uint8)
    for v in cpu_ref.process_yaw_and_pitchs(pano, 33, [30, 90, 150], 40, 30, 100):
        assert (v == np.array([10, 128, 250], np.uint8)).all()


def test_view_centre_reads_three_quarters_across(synth):
    pano = np.zeros((512, 1024, 3), np.uint8)
    pano[:, :, 0] = (np.arange(1024) // 4).astype(np.uint8)[None, :]
    pano[:, :, 1] = (np.arange(512) // 2).astype(np.uint8)[:, None]
    v = cpu_ref.process_yaw_and_pitchs(pano, 0, [90, 45], 64, 64, 90)
    assert v[0][32, 32].tolist()[:2] == [768 // 4, 256 // 2]  # (3*pw/4, ph*90/180)
    assert v[1][32, 32].tolist()[:2] == [768 // 4, 128 // 2]  # pitch is the polar angle from the zenith


def test_legacy_entry_point_reflect_equals_constant_for_clipped_maps(synth):
    pano = synth.synth_pano(256, 128, 1003, "N")
    U, V = maps.pitch_map_deg(64, 48, 70, 256, 128, 90)
    assert np.array_equal(cpu_ref.panorama_to_plane(pano, U, V), cpu_ref.remap(pano, U, V, cpu_ref.BORDER_CONSTANT))
