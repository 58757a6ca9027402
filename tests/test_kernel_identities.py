"""Arithmetic identities the HIP kernels lean on, checked exhaustively on the CPU (no GPU, no library): the
packed forms must give cv::remap's fixed-point result for EVERY input, not only for the ones a parity run meets.

Stage 2 (p2p_views.hip, blend4_packed / tap_weights): cv::remap's bilinear weights for 5-bit fractions are
w = 32 * [(32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx*fy] with (sum + 2^14) >> 15 (oracle/cv_remap_oracle.c's
table, read back here through cpu_ref.weight_table(); at fx = fy = 0 it is [32767, 0, 0, 1]); the kernel uses w' = 64 * [(32-fx)(32-fy), ...] as u16 and takes byte 2 of sum' + 2^15.  The one product that does
not fit 16 bits (fx = fy = 0: 65536) is stored as 32 * 2047 = 65504.
Stage 1 (rot_blend / vmad24): (32-f) a + f b + 16 >> 5 computed as the high byte of a 16-bit field holding
8 (32-f) a + 8 f b + 128.
"""
import numpy as np

from oracle import cpu_ref


def test_stage2_u16_weights_give_cv_remap_rounding_for_every_fraction():
    rng = np.random.default_rng(5)
    taps = np.concatenate([rng.integers(0, 256, size=(4000, 4)), [[0, 0, 0, 0], [255, 255, 255, 255], [255, 0, 0, 255],
                                                                 [0, 255, 255, 0], [1, 254, 3, 252]]]).astype(np.int64)
    wtab = np.asarray(cpu_ref.weight_table(), dtype=np.int64).reshape(32, 32, 4)  # [fy][fx][4], the oracle's table
    for fx in range(32):
        for fy in range(32):
            gx, gy = 32 - fx, 32 - fy
            w_cv = wtab[fy, fx]
            if fx or fy:  # the table is the product form; at (0, 0) OpenCV's short saturates: [32767, 0, 0, 1]
                assert np.array_equal(w_cv, 32 * np.array([gx * gy, fx * gy, gx * fy, fx * fy])), (fx, fy)
            want = (taps @ w_cv + (1 << 14)) >> 15  # INTER_REMAP_COEF_SCALE = 2^15
            up = 2048 - 64 * fy - (1 if (fx | fy) == 0 else 0)
            lo = 64 * fy
            w_k = np.array([gx * up, fx * up, gx * lo, fx * lo], dtype=np.int64)
            assert w_k.max() <= 0xFFFF, (fx, fy)                 # every weight is a u16
            acc = taps @ w_k + 32768
            assert acc.max() < 1 << 32                            # v_dot2_u32_u16 accumulates in 32 bits
            got = (acc >> 16) & 0xFF
            assert np.array_equal(got, want), (fx, fy)


def test_stage2_copy_weight_is_exact_for_every_byte():
    a = np.arange(256, dtype=np.int64)
    assert np.array_equal((a * 65504 + 32768) >> 16, a)


def test_stage1_high_byte_form_is_the_rounded_shift_for_every_input():
    a, b = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    for f in range(33):
        want = ((32 - f) * a + f * b + 16) >> 5
        field = 8 * (32 - f) * a + 8 * f * b + 128
        assert field.max() <= 0xFFFF
        assert np.array_equal(field >> 8, want), f
