// p2p_float.hip -- opt-in float pixel path (not in the reference)
//   not in the reference (opt-in): float_views_kernel, one float32 / float16 resample per view with wrap-around
// Reference behaviour (cited, never copied):
//   P = /root/reference/app/panorama_to_plane-pitch.py, L = /root/reference/app/legacy/panorama_to_plane.py
// The fixed-point arithmetic is OpenCV 4.10's (imgwarp.cpp remapBilinear, INTER_BITS = 5,
// INTER_REMAP_COEF_BITS = 15); see DESIGN.md "Arithmetic contract".
// Compiled with -ffp-contract=off: every float operation below rounds where NumPy rounds.
#include <hip/hip_fp16.h>
#include "p2p_inline.h"

namespace p2p {

// ---------------------------------------------------------------------------------------------
// Float pixel path (opt-in, BEYOND the reference: BASELINE config 5's "fp16 pixel path" and SURVEY 8(f)4's
// quality mode).  One resample instead of two: the pitch map's float coordinate is shifted by the yaw's
// column offset yaw * pw / 360 with true wrap-around at the seam, nothing is quantised to 1/32 pixel and no
// intermediate image is rounded to uint8; the 2x2 blend runs in float32 or in packed float16.  The result
// is rounded to uint8 once.  Not bit-comparable with cv2.remap by construction; tests bound it against a
// float32 NumPy evaluation of the same formula and against the exact path on band-limited panoramas.
// ---------------------------------------------------------------------------------------------
template <bool HALF>
__global__ __launch_bounds__(256) void float_views_kernel(ViewsParams P, const double* __restrict__ yaw_rad)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int pitch_i = blockIdx.z;
    if (x >= P.ow || y >= P.oh)
        return;
    float U, V;
    const PitchConst pc = P.pitch[pitch_i];
    // P.centre = 0: the reference's convention (integer coordinates are sample points, P:122-131);
    // 0.5 (P2P_FLAG_PIXEL_CENTRES): rays through output-pixel centres, panorama texel i centred at i + 0.5
    pitch_map_eval((float)x + P.centre, (float)y + P.centre, P.geom, pc.c, pc.s, U, V);
    const bool dead = !(U == U) || !(V == V);  // NaN next to a pole: black, as in the exact path
    V -= P.centre;
    if (V < 0.0f)
        V = 0.0f;
    const int y0 = dead ? 0 : (int)V;          // V in [0, ph - 1]
    const float wy = dead ? 0.0f : V - (float)y0;
    const int y1 = y0 + 1 < P.ph ? y0 + 1 : y0;
    const size_t view_bytes = (size_t)P.oh * P.ow * 3;
    const size_t px_off = ((size_t)y * P.ow + x) * 3;
    for (int pano = 0; pano < P.n_panos; ++pano) {
        const uint8_t* __restrict__ S = P.src + (size_t)pano * P.pano_stride;
        const uint8_t* __restrict__ r0 = S + (size_t)y0 * P.src_pitch;
        const uint8_t* __restrict__ r1 = S + (size_t)y1 * P.src_pitch;
        for (int yi = 0; yi < P.n_yaw; ++yi) {
            uint8_t* O = P.out + (((size_t)pano * P.n_yaw + yi) * P.n_pitch + pitch_i) * view_bytes + px_off;
            if (dead) {
                O[0] = O[1] = O[2] = 0;
                continue;
            }
            // source column = U + yaw * pw / 2 pi (mod pw): P:98-101 without the clip at the seam
            double sh = fmod(yaw_rad[yi] * (double)P.pw / 6.283185307179586, (double)P.pw);
            if (sh < 0.0)
                sh += (double)P.pw;
            float xs = U + (float)sh - P.centre;
            if (xs < 0.0f)
                xs += (float)P.pw;
            if (xs >= (float)P.pw)
                xs -= (float)P.pw;
            int x0 = (int)xs;
            if (x0 >= P.pw)
                x0 = P.pw - 1;
            const float wx = xs - (float)x0;
            const int x1 = x0 + 1 < P.pw ? x0 + 1 : 0;  // wrap-around
            uint32_t a, b, c, d;
            {
                uint2 q0, q1;
                __builtin_memcpy(&q0, r0 + 3 * x0, 8);
                __builtin_memcpy(&q1, r1 + 3 * x0, 8);
                a = q0.x;
                c = q1.x;
                b = __builtin_amdgcn_alignbyte(q0.y, q0.x, 3);
                d = __builtin_amdgcn_alignbyte(q1.y, q1.x, 3);
                if (x1 == 0) {
                    __builtin_memcpy(&b, r0, 4);
                    __builtin_memcpy(&d, r1, 4);
                }
            }
            uint32_t res = 0;
            if (HALF) {
                const __half2 hx = __float2half2_rn(wx), hy = __float2half2_rn(wy);
                auto pair = [](uint32_t p, int s0, int s1) {
                    return __halves2half2(__ushort2half_rn((unsigned short)((p >> s0) & 0xFFu)),
                                          __ushort2half_rn((unsigned short)((p >> s1) & 0xFFu)));
                };
#pragma unroll
                for (int k = 0; k < 2; ++k) {  // (B, G) then (R, R)
                    const int s0 = k ? 16 : 0, s1 = k ? 16 : 8;
                    const __half2 pa = pair(a, s0, s1), pb = pair(b, s0, s1), pc2 = pair(c, s0, s1), pd = pair(d, s0, s1);
                    const __half2 h0 = __hfma2(hx, __hsub2(pb, pa), pa);
                    const __half2 h1 = __hfma2(hx, __hsub2(pd, pc2), pc2);
                    const __half2 v = __hfma2(hy, __hsub2(h1, h0), h0);
                    int lo = __half2int_rn(__low2half(v)), hi = __half2int_rn(__high2half(v));
                    lo = lo < 0 ? 0 : (lo > 255 ? 255 : lo);
                    hi = hi < 0 ? 0 : (hi > 255 ? 255 : hi);
                    res |= k ? (uint32_t)lo << 16 : ((uint32_t)lo | (uint32_t)hi << 8);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float pa = (float)((a >> (8 * k)) & 0xFFu), pb = (float)((b >> (8 * k)) & 0xFFu);
                    const float pc2 = (float)((c >> (8 * k)) & 0xFFu), pd = (float)((d >> (8 * k)) & 0xFFu);
                    const float h0 = __builtin_fmaf(wx, pb - pa, pa);
                    const float h1 = __builtin_fmaf(wx, pd - pc2, pc2);
                    const float v = __builtin_fmaf(wy, h1 - h0, h0);
                    int r = (int)__builtin_rintf(v);
                    r = r < 0 ? 0 : (r > 255 ? 255 : r);
                    res |= (uint32_t)r << (8 * k);
                }
            }
            O[0] = (uint8_t)res;
            O[1] = (uint8_t)(res >> 8);
            O[2] = (uint8_t)(res >> 16);
        }
    }
}

hipError_t launch_float_views(const ViewsParams& P, const double* yaw_rad, bool half, hipStream_t st)
{
    dim3 grid((P.ow + 63) / 64, (P.oh + 3) / 4, P.n_pitch);
    if (half)
        hipLaunchKernelGGL(float_views_kernel<true>, grid, dim3(256), 0, st, P, yaw_rad);
    else
        hipLaunchKernelGGL(float_views_kernel<false>, grid, dim3(256), 0, st, P, yaw_rad);
    return hipGetLastError();
}

}  // namespace p2p
