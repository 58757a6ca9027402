/*
 * oracle/cv_remap_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the arithmetic the reference delegates to its third-party
 * dependency opencv-python==4.10.0.84 (pinned in /root/reference/pyproject.toml:11,
 * requirements.txt:13, uv.lock:119-120) at these call sites:
 *
 *   app/panorama_to_plane-pitch.py:192-199  cv2.remap(pano, U_yaw, V_yaw, INTER_LINEAR, BORDER_CONSTANT)
 *   app/panorama_to_plane-pitch.py:212-218  cv2.remap(rot,  U_p,   V_p,   INTER_LINEAR, BORDER_CONSTANT)
 *   app/legacy/panorama_to_plane.py:179     cv2.remap(img,  U,     V,     interp,       BORDER_REFLECT)
 *
 * OpenCV's source is NOT under /root/reference and cv2 is not installable in the
 * build container (no wheel, no network), and the reference ships no tests or
 * golden images.  This file therefore restates the PUBLISHED algorithm of
 * OpenCV 4.10 modules/imgproc/src/imgwarp.cpp (cv::remap -> RemapInvoker ->
 * remapBilinear<FixedPtCast<int,uchar,15>, RemapVec_8u, short>, initInterTab2D,
 * cv::borderInterpolate) for CV_8U sources with two CV_32FC1 maps:
 *
 *   PARITY UNPINNED for the gather: no reference-held vector exists to check this
 *   restatement against.  (The float coordinate maps ARE pinned: tests/golden/ holds
 *   vectors produced by importing the reference's own map builders.)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (libp2p_hip.so) never links or calls it.
 *
 * Restated semantics (x86-64 build of OpenCV, which is what the pinned wheel is):
 *  1. sx = cvRound(mapx * 32), sy = cvRound(mapy * 32): float32 multiply, then
 *     cvtss2si (round-half-even; NaN / out-of-int32-range -> 0x80000000).
 *  2. ix = saturate_cast<short>(sx >> 5), fx = sx & 31 (same for y).
 *  3. weights: 32x32 table of 4 shorts built exactly as initInterTab2D does
 *     (float products, saturate_cast<short>(v * 32768), sum fix-up), which gives
 *     {32(32-fx)(32-fy), 32 fx (32-fy), 32 (32-fx) fy, 32 fx fy} except the
 *     (0,0) cell which becomes {32767, 0, 0, 1}.
 *  4. out = saturate_u8((w0*p00 + w1*p01 + w2*p10 + w3*p11 + (1 << 14)) >> 15).
 *  5. border: taps outside the image are resolved by borderInterpolate();
 *     BORDER_CONSTANT taps read borderValue; a pixel whose 2x2 footprint is
 *     entirely outside gets borderValue unblended.
 *  6. The SIMD (RemapVec_8u) path is integer arithmetic identical to the scalar
 *     one; IPP is compiled out for remap (IPP_DISABLE_REMAP).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>

#define ORC_INTER_BITS 5
#define ORC_INTER_TAB_SIZE 32
#define ORC_COEF_BITS 15
#define ORC_COEF_SCALE (1 << ORC_COEF_BITS)

/* OpenCV border codes (core/base.hpp) */
enum { ORC_BORDER_CONSTANT = 0, ORC_BORDER_REPLICATE = 1, ORC_BORDER_REFLECT = 2,
       ORC_BORDER_WRAP = 3, ORC_BORDER_REFLECT_101 = 4 };

static short g_wtab[ORC_INTER_TAB_SIZE * ORC_INTER_TAB_SIZE + 2][4];
static int g_wtab_ready = 0;

/* cvRound(float) on x86-64: _mm_cvtss_si32 */
static inline int orc_cvround_f32(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) /* NaN or out of range */
        return INT_MIN;
    return (int)nearbyintf(v); /* default FP environment: round-half-even */
}

static inline short orc_sat_short(int v)
{
    return (short)(v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v));
}

static inline short orc_sat_short_f(float v)
{
    int iv = orc_cvround_f32(v);
    return orc_sat_short(iv);
}

/* initInterTab2D(INTER_LINEAR, fixpt=true), imgwarp.cpp */
static void orc_init_wtab(void)
{
    float tab1d[ORC_INTER_TAB_SIZE * 2];
    const float scale = 1.f / ORC_INTER_TAB_SIZE;
    int i, j, k1, k2;
    const int ksize = 2;
    memset(g_wtab, 0, sizeof(g_wtab));
    for (i = 0; i < ORC_INTER_TAB_SIZE; i++) { /* interpolateLinear */
        float x = i * scale;
        tab1d[i * 2 + 0] = 1.f - x;
        tab1d[i * 2 + 1] = x;
    }
    for (i = 0; i < ORC_INTER_TAB_SIZE; i++)
        for (j = 0; j < ORC_INTER_TAB_SIZE; j++) {
            short* itab = &g_wtab[i * ORC_INTER_TAB_SIZE + j][0];
            int isum = 0;
            for (k1 = 0; k1 < ksize; k1++) {
                float vy = tab1d[i * ksize + k1];
                for (k2 = 0; k2 < ksize; k2++) {
                    float v = vy * tab1d[j * ksize + k2];
                    itab[k1 * ksize + k2] = orc_sat_short_f(v * ORC_COEF_SCALE);
                    isum += itab[k1 * ksize + k2];
                }
            }
            if (isum != ORC_COEF_SCALE) {
                /* literal restatement, including the ksize/2-based scan window that
                   for ksize == 2 looks at itab[3..6] (the next cell is still zero). */
                int diff = isum - ORC_COEF_SCALE;
                int ksize2 = ksize / 2, Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (k1 = ksize2; k1 < ksize2 + 2; k1++)
                    for (k2 = ksize2; k2 < ksize2 + 2; k2++) {
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2])
                            mk1 = k1, mk2 = k2;
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2])
                            Mk1 = k1, Mk2 = k2;
                    }
                if (diff < 0)
                    itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
                else
                    itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
            }
        }
    g_wtab_ready = 1;
}

/* cv::borderInterpolate, core/src/copy.cpp */
static int orc_border_interpolate(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (border == ORC_BORDER_REPLICATE)
        return p < 0 ? 0 : len - 1;
    if (border == ORC_BORDER_REFLECT || border == ORC_BORDER_REFLECT_101) {
        int delta = border == ORC_BORDER_REFLECT_101;
        if (len == 1)
            return 0;
        do {
            if (p < 0)
                p = -p - 1 + delta;
            else
                p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (border == ORC_BORDER_WRAP) {
        if (p < 0)
            p -= ((p - len + 1) / len) * len;
        if (p >= len)
            p %= len;
        return p;
    }
    return -1; /* BORDER_CONSTANT */
}

static inline uint8_t orc_fixedpt_cast_u8(int v)
{
    v = (v + (1 << (ORC_COEF_BITS - 1))) >> ORC_COEF_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* Export the weight table so tests can check it against the closed form. */
void orc_get_wtab(short* out /* [1024][4] */)
{
    if (!g_wtab_ready)
        orc_init_wtab();
    memcpy(out, g_wtab, sizeof(short) * 4 * ORC_INTER_TAB_SIZE * ORC_INTER_TAB_SIZE);
}

/* Quantise float maps the way RemapInvoker does (planar CV_32FC1 maps).
   ixy: [n][2] int16 (x, y), fxy: [n] uint16 = fy*32 + fx. */
void orc_quantise_maps(const float* mapx, const float* mapy, int64_t n, int16_t* ixy, uint16_t* fxy)
{
    for (int64_t k = 0; k < n; k++) {
        int sx = orc_cvround_f32(mapx[k] * (float)ORC_INTER_TAB_SIZE);
        int sy = orc_cvround_f32(mapy[k] * (float)ORC_INTER_TAB_SIZE);
        ixy[2 * k + 0] = orc_sat_short(sx >> ORC_INTER_BITS);
        ixy[2 * k + 1] = orc_sat_short(sy >> ORC_INTER_BITS);
        fxy[k] = (uint16_t)((sy & (ORC_INTER_TAB_SIZE - 1)) * ORC_INTER_TAB_SIZE +
                            (sx & (ORC_INTER_TAB_SIZE - 1)));
    }
}

/*
 * cv::remap(src, dst, mapx, mapy, INTER_LINEAR, border, borderValue) for CV_8UC(cn).
 * Strides are in bytes for images and in elements (floats) for maps.
 * Returns 0 on success, negative on invalid arguments (OpenCV would CV_Assert).
 */
int orc_remap_u8(const uint8_t* src, int sw, int sh, int64_t sstride, int cn,
                 const float* mapx, const float* mapy, int64_t mstride,
                 uint8_t* dst, int dw, int dh, int64_t dstride,
                 int border, const uint8_t* border_value /* cn bytes or NULL = zeros */)
{
    uint8_t cval[4] = {0, 0, 0, 0};
    if (!src || !mapx || !mapy || !dst || cn < 1 || cn > 4)
        return -1;
    /* CV_Assert(dst.cols < SHRT_MAX && dst.rows < SHRT_MAX && src.cols < SHRT_MAX && src.rows < SHRT_MAX) */
    if (!(dw < SHRT_MAX && dh < SHRT_MAX && sw < SHRT_MAX && sh < SHRT_MAX) || sw <= 0 || sh <= 0)
        return -2;
    if (border < ORC_BORDER_CONSTANT || border > ORC_BORDER_REFLECT_101)
        return -3;
    if (border_value)
        memcpy(cval, border_value, (size_t)cn);
    if (!g_wtab_ready)
        orc_init_wtab();

    const unsigned width1 = (unsigned)(sw - 1 > 0 ? sw - 1 : 0);
    const unsigned height1 = (unsigned)(sh - 1 > 0 ? sh - 1 : 0);

    for (int dy = 0; dy < dh; dy++) {
        const float* mx = mapx + (int64_t)dy * mstride;
        const float* my = mapy + (int64_t)dy * mstride;
        uint8_t* D = dst + (int64_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++, D += cn) {
            int qx = orc_cvround_f32(mx[dx] * (float)ORC_INTER_TAB_SIZE);
            int qy = orc_cvround_f32(my[dx] * (float)ORC_INTER_TAB_SIZE);
            int sx = orc_sat_short(qx >> ORC_INTER_BITS);
            int sy = orc_sat_short(qy >> ORC_INTER_BITS);
            const short* w = g_wtab[(qy & 31) * ORC_INTER_TAB_SIZE + (qx & 31)];
            if ((unsigned)sx < width1 && (unsigned)sy < height1) {
                /* inlier run of remapBilinear: all four taps inside */
                const uint8_t* S = src + (int64_t)sy * sstride + (int64_t)sx * cn;
                for (int k = 0; k < cn; k++)
                    D[k] = orc_fixedpt_cast_u8(S[k] * w[0] + S[k + cn] * w[1] +
                                               S[sstride + k] * w[2] + S[sstride + k + cn] * w[3]);
                continue;
            }
            if (border == ORC_BORDER_CONSTANT &&
                (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) {
                for (int k = 0; k < cn; k++)
                    D[k] = cval[k];
                continue;
            }
            int sx0, sx1, sy0, sy1;
            const uint8_t *v0, *v1, *v2, *v3;
            if (border == ORC_BORDER_REPLICATE) {
                sx0 = sx < 0 ? 0 : (sx > sw - 1 ? sw - 1 : sx);
                sx1 = sx + 1 < 0 ? 0 : (sx + 1 > sw - 1 ? sw - 1 : sx + 1);
                sy0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
                sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
                v0 = src + (int64_t)sy0 * sstride + (int64_t)sx0 * cn;
                v1 = src + (int64_t)sy0 * sstride + (int64_t)sx1 * cn;
                v2 = src + (int64_t)sy1 * sstride + (int64_t)sx0 * cn;
                v3 = src + (int64_t)sy1 * sstride + (int64_t)sx1 * cn;
            } else {
                sx0 = orc_border_interpolate(sx, sw, border);
                sx1 = orc_border_interpolate(sx + 1, sw, border);
                sy0 = orc_border_interpolate(sy, sh, border);
                sy1 = orc_border_interpolate(sy + 1, sh, border);
                v0 = sx0 >= 0 && sy0 >= 0 ? src + (int64_t)sy0 * sstride + (int64_t)sx0 * cn : cval;
                v1 = sx1 >= 0 && sy0 >= 0 ? src + (int64_t)sy0 * sstride + (int64_t)sx1 * cn : cval;
                v2 = sx0 >= 0 && sy1 >= 0 ? src + (int64_t)sy1 * sstride + (int64_t)sx0 * cn : cval;
                v3 = sx1 >= 0 && sy1 >= 0 ? src + (int64_t)sy1 * sstride + (int64_t)sx1 * cn : cval;
            }
            for (int k = 0; k < cn; k++)
                D[k] = orc_fixedpt_cast_u8(v0[k] * w[0] + v1[k] * w[1] + v2[k] * w[2] + v3[k] * w[3]);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * INTER_NEAREST and INTER_CUBIC: the other two entries of the legacy tool's method table
 * (app/legacy/panorama_to_plane.py:172-176; only 'bilinear' is reachable from its own callers, L:194).
 * Same status as above: restated from OpenCV 4.10 imgwarp.cpp (remapNearest, remapBicubic,
 * initInterTab1D/2D with interpolateCubic), PARITY UNPINNED.
 * ------------------------------------------------------------------------------------------------ */

/* RemapInvoker, nearest branch with two CV_32FC1 maps: XY = saturate_cast<short>(map) (cvRound, then
 * saturation); remapNearest copies the pixel or resolves the border. */
int orc_remap_nearest_u8(const uint8_t* src, int sw, int sh, int64_t sstride, int cn,
                         const float* mapx, const float* mapy, int64_t mstride,
                         uint8_t* dst, int dw, int dh, int64_t dstride,
                         int border, const uint8_t* border_value)
{
    uint8_t cval[4] = {0, 0, 0, 0};
    if (!src || !mapx || !mapy || !dst || cn < 1 || cn > 4)
        return -1;
    if (!(dw < SHRT_MAX && dh < SHRT_MAX && sw < SHRT_MAX && sh < SHRT_MAX) || sw <= 0 || sh <= 0)
        return -2;
    if (border < ORC_BORDER_CONSTANT || border > ORC_BORDER_REFLECT_101)
        return -3;
    if (border_value)
        memcpy(cval, border_value, (size_t)cn);
    for (int dy = 0; dy < dh; dy++) {
        const float* mx = mapx + (int64_t)dy * mstride;
        const float* my = mapy + (int64_t)dy * mstride;
        uint8_t* D = dst + (int64_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++, D += cn) {
            int sx = orc_sat_short_f(mx[dx]), sy = orc_sat_short_f(my[dx]);
            const uint8_t* S;
            if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) {
                S = src + (int64_t)sy * sstride + (int64_t)sx * cn;
            } else if (border == ORC_BORDER_REPLICATE) {
                sx = sx < 0 ? 0 : (sx > sw - 1 ? sw - 1 : sx);
                sy = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
                S = src + (int64_t)sy * sstride + (int64_t)sx * cn;
            } else if (border == ORC_BORDER_CONSTANT) {
                S = cval;
            } else {
                sx = orc_border_interpolate(sx, sw, border);
                sy = orc_border_interpolate(sy, sh, border);
                S = src + (int64_t)sy * sstride + (int64_t)sx * cn;
            }
            for (int k = 0; k < cn; k++)
                D[k] = S[k];
        }
    }
    return 0;
}

static short g_ctab[ORC_INTER_TAB_SIZE * ORC_INTER_TAB_SIZE][16];
static int g_ctab_ready = 0;

/* interpolateCubic (A = -0.75) in float, initInterTab1D / initInterTab2D(INTER_CUBIC, fixpt) */
static void orc_init_ctab(void)
{
    float tab1d[ORC_INTER_TAB_SIZE * 4];
    const float scale = 1.f / ORC_INTER_TAB_SIZE;
    const int ksize = 4;
    int i, j, k1, k2;
    for (i = 0; i < ORC_INTER_TAB_SIZE; i++) {
        const float A = -0.75f;
        float x = i * scale;
        float* c = tab1d + i * 4;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    }
    for (i = 0; i < ORC_INTER_TAB_SIZE; i++)
        for (j = 0; j < ORC_INTER_TAB_SIZE; j++) {
            short* itab = g_ctab[i * ORC_INTER_TAB_SIZE + j];
            int isum = 0;
            for (k1 = 0; k1 < ksize; k1++) {
                float vy = tab1d[i * ksize + k1];
                for (k2 = 0; k2 < ksize; k2++) {
                    float v = vy * tab1d[j * ksize + k2];
                    itab[k1 * ksize + k2] = orc_sat_short_f(v * ORC_COEF_SCALE);
                    isum += itab[k1 * ksize + k2];
                }
            }
            if (isum != ORC_COEF_SCALE) {
                int diff = isum - ORC_COEF_SCALE;
                int ksize2 = ksize / 2, Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (k1 = ksize2; k1 < ksize2 + 2; k1++)
                    for (k2 = ksize2; k2 < ksize2 + 2; k2++) {
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2])
                            mk1 = k1, mk2 = k2;
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2])
                            Mk1 = k1, Mk2 = k2;
                    }
                if (diff < 0)
                    itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
                else
                    itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
            }
        }
    g_ctab_ready = 1;
}

void orc_get_cubic_wtab(short* out /* [1024][16] */)
{
    if (!g_ctab_ready)
        orc_init_ctab();
    memcpy(out, g_ctab, sizeof(g_ctab));
}

/* remapBicubic<FixedPtCast<int,uchar,15>, short, INTER_REMAP_COEF_SCALE> */
int orc_remap_cubic_u8(const uint8_t* src, int sw, int sh, int64_t sstride, int cn,
                       const float* mapx, const float* mapy, int64_t mstride,
                       uint8_t* dst, int dw, int dh, int64_t dstride,
                       int border, const uint8_t* border_value)
{
    uint8_t cval[4] = {0, 0, 0, 0};
    if (!src || !mapx || !mapy || !dst || cn < 1 || cn > 4)
        return -1;
    if (!(dw < SHRT_MAX && dh < SHRT_MAX && sw < SHRT_MAX && sh < SHRT_MAX) || sw <= 0 || sh <= 0)
        return -2;
    if (border < ORC_BORDER_CONSTANT || border > ORC_BORDER_REFLECT_101)
        return -3;
    if (border_value)
        memcpy(cval, border_value, (size_t)cn);
    if (!g_ctab_ready)
        orc_init_ctab();
    const unsigned width1 = (unsigned)(sw - 3 > 0 ? sw - 3 : 0);
    const unsigned height1 = (unsigned)(sh - 3 > 0 ? sh - 3 : 0);
    for (int dy = 0; dy < dh; dy++) {
        const float* mx = mapx + (int64_t)dy * mstride;
        const float* my = mapy + (int64_t)dy * mstride;
        uint8_t* D = dst + (int64_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++, D += cn) {
            int qx = orc_cvround_f32(mx[dx] * (float)ORC_INTER_TAB_SIZE);
            int qy = orc_cvround_f32(my[dx] * (float)ORC_INTER_TAB_SIZE);
            int sx = orc_sat_short(qx >> ORC_INTER_BITS) - 1;
            int sy = orc_sat_short(qy >> ORC_INTER_BITS) - 1;
            const short* w = g_ctab[(qy & 31) * ORC_INTER_TAB_SIZE + (qx & 31)];
            if ((unsigned)sx < width1 && (unsigned)sy < height1) {
                const uint8_t* S = src + (int64_t)sy * sstride + (int64_t)sx * cn;
                for (int k = 0; k < cn; k++) {
                    int sum = 0;
                    for (int r = 0; r < 4; r++)
                        for (int c = 0; c < 4; c++)
                            sum += S[(int64_t)r * sstride + c * cn + k] * w[r * 4 + c];
                    D[k] = orc_fixedpt_cast_u8(sum);
                }
                continue;
            }
            if (border == ORC_BORDER_CONSTANT && (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0)) {
                for (int k = 0; k < cn; k++)
                    D[k] = cval[k];
                continue;
            }
            int x[4], y[4];
            for (int i = 0; i < 4; i++) {
                x[i] = orc_border_interpolate(sx + i, sw, border);
                y[i] = orc_border_interpolate(sy + i, sh, border);
            }
            for (int k = 0; k < cn; k++) {
                int cv = cval[k], sum = cv * ORC_COEF_SCALE;
                for (int r = 0; r < 4; r++) {
                    if (y[r] < 0)
                        continue;
                    const uint8_t* S = src + (int64_t)y[r] * sstride;
                    for (int c = 0; c < 4; c++)
                        if (x[c] >= 0)
                            sum += (S[x[c] * cn + k] - cv) * w[r * 4 + c];
                }
                D[k] = orc_fixedpt_cast_u8(sum);
            }
        }
    }
    return 0;
}
