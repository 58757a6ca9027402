// p2p_inline.h -- small device functions shared by the kernel translation units (p2p_maps.hip, p2p_views.hip,
// p2p_remap.hip, p2p_float.hip).  Compiled with -ffp-contract=off: every float operation rounds where NumPy rounds.
#ifndef P2P_INLINE_H
#define P2P_INLINE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "p2p_device.h"

namespace p2p {

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_round_f32(float v)
{
    // cvRound(float) on x86-64 = cvtss2si: round-half-even, NaN / out of range -> INT_MIN
    if (!(v >= -2147483648.0f && v < 2147483648.0f))
        return INT32_MIN;
    return (int)__builtin_rintf(v);
}

__device__ __forceinline__ int sat_short(int v)
{
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

__device__ __forceinline__ float clip_keep_nan(float v, float lo, float hi)
{
    // np.clip propagates NaN; fminf/fmaxf would not
    return v < lo ? lo : (v > hi ? hi : v);
}

// P:114-175 for one output pixel, float32 throughout, same operation order as NumPy:
//   x**2 + y**2 + z**2 left to right, IEEE sqrt and divide, the 3x3 float32 sgemm as the
//   sequential-FMA accumulation OpenBLAS performs (acc = fma(R[i][k], v[k], acc), k = 0..2),
//   arccos / arctan2 % 2pi, scale, clip.
// clip_u = false (float pixel path only): U stays the unclipped azimuth in [0, pw) -- that path wraps around the
// seam instead of collapsing (pw - 1, pw) onto column pw - 1 as P:172 does.
__device__ __forceinline__ void pitch_map_eval(float u, float v, const MapGeom& g, float c, float s,
                                               float& U, float& V, bool clip_u = true)
{
    const float TWO_PI_F = 6.283185307179586f;  // float32(2*np.pi), the "weak" Python scalar
    const float PI_F = 3.141592653589793f;
    float x = u - g.half_w;  // P:129
    float y = g.half_h - v;  // P:130
    float z = g.focal;       // P:131
    float n = __fsqrt_rn(x * x + y * y + z * z);  // P:134
    // P:137-139: x/n, y/n, z/n.  One IEEE reciprocal, then q = a*r corrected by its exact FMA
    // residual: RN(q + (a - q*n)*r) is the correctly rounded quotient (Markstein) -- the same bits as
    // three IEEE divisions at half their cost.
    const float r = __fdiv_rn(1.0f, n);
    float xn = x * r, yn = y * r, zn = z * r;
    xn = __builtin_fmaf(__builtin_fmaf(-xn, n, x), r, xn);
    yn = __builtin_fmaf(__builtin_fmaf(-yn, n, y), r, yn);
    zn = __builtin_fmaf(__builtin_fmaf(-zn, n, z), r, zn);
    float yr = __builtin_fmaf(-s, zn, c * yn);  // P:155 row 1: [0, cos, -sin]
    float zr = __builtin_fmaf(c, zn, s * yn);   // P:155 row 2: [0, sin,  cos]
    float theta = acosf(zr);                    // P:162 (NaN if zr rounds above 1)
    float phi = atan2f(yr, xn);                 // P:164
    if (phi < 0.0f)
        phi += TWO_PI_F;  // floored '%': |phi| <= pi so fmod is the identity; -0.0 -> +0.0 either way
    else if (phi == 0.0f)
        phi = 0.0f;
    // P:167 / P:169: division by the constants float32(2 pi) / float32(pi), same correction scheme
    const float R_TWO_PI = 0.15915494f, R_PI = 0.31830987f;  // RN(1 / 6.2831855f), RN(1 / 3.1415927f)
    const float tu = phi * g.pw_f, tv = theta * g.ph_f;
    U = tu * R_TWO_PI;
    V = tv * R_PI;
    U = __builtin_fmaf(__builtin_fmaf(-U, TWO_PI_F, tu), R_TWO_PI, U);
    V = __builtin_fmaf(__builtin_fmaf(-V, PI_F, tv), R_PI, V);
    if (clip_u)
        U = clip_keep_nan(U, 0.0f, g.pw_f - 1.0f);  // P:172
    else if (U >= g.pw_f)
        U -= g.pw_f;
    V = clip_keep_nan(V, 0.0f, g.ph_f - 1.0f);  // P:173
}

// cv::borderInterpolate (core/src/copy.cpp) for the border codes of include/p2p_hip.h
__device__ __forceinline__ int border_interpolate(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (border == 1)  // REPLICATE
        return p < 0 ? 0 : len - 1;
    if (border == 2 || border == 4) {  // REFLECT / REFLECT_101
        int delta = border == 4;
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
    if (border == 3) {  // WRAP
        if (p < 0)
            p -= ((p - len + 1) / len) * len;
        if (p >= len)
            p %= len;
        return p;
    }
    return -1;  // CONSTANT
}

}  // namespace p2p
#endif  // P2P_INLINE_H
