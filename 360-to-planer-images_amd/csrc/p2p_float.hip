// p2p_float.hip -- opt-in float pixel path (not in the reference): float_views_kernel, float_views_rest_kernel
// Reference behaviour (cited, never copied):
//   P = /root/reference/app/panorama_to_plane-pitch.py
// BASELINE config 5's "fp16 pixel path" and SURVEY 8(f)4's quality mode.  One resample instead of two: the pitch
// map's float coordinate (P:114-175, evaluated by the plan pass with the azimuth left unclipped) is shifted by the
// yaw's column offset yaw * pw / 360 (P:98-101 without the clip) with true wrap-around at the seam, nothing is
// quantised to 1/32 pixel and no intermediate image is rounded to uint8; the 2x2 blend runs in float32 or in
// packed float16 and is rounded to uint8 once (round-half-even, as np.rint).  Not bit-comparable with cv2.remap by
// construction; tests bound it against a float32 NumPy evaluation of the same formula and against the exact path
// on band-limited panoramas.
//
// Same machinery as the exact kernel (p2p_views.hip): per-job plan (piece headers, per-row footprint spans, LDS tap
// offsets per pixel -- here of floor(U), floor(V) -- plus the coordinate fractions in 1/65536), source rows staged
// through LDS as 16-byte pieces (stage 1 is a plain copy: the yaw's INTEGER column shift picks the pieces, its
// fraction is added to the pixel's own per yaw and may carry the taps one column on), one barrier per yaw, the
// pixels leave as 12-byte non-temporal buffer stores.  Stage 1 is cheaper than the exact path's (a copy), the
// per-pixel part is not: the float blend, the per-yaw weights and the carry come to about 100 issue cycles per pixel
// against 64 (byte <-> float conversions and dot products issue at the same rate as the packed integer multiplies),
// so the two paths run at the same speed.  Tiles the plan marks for gathers (view seam, pole, widths not divisible by 4)
// are drawn by float_views_rest_kernel, one thread per pixel from global memory.
#include <hip/hip_fp16.h>
#include <type_traits>
#include "p2p_tile.h"

namespace p2p {
namespace P2P_SHAPE_NS {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// yaw shift in source columns, [0, pw): P:98-101 without the clip at the seam
__device__ __forceinline__ double yaw_shift(double yaw_rad, int pw)
{
    double sh = fmod(yaw_rad * (double)pw / 6.283185307179586, (double)pw);
    if (sh < 0.0)
        sh += (double)pw;
    if (sh >= (double)pw)
        sh = 0.0;
    return sh;
}

// 2x2 blend of four BGRx taps, float32: per channel h0 = a + wx (b - a), h1 = c + wx (d - c), v = h0 + wy (h1 - h0),
// rounded half-even and clamped by v_cvt_pk_u8_f32
__device__ __forceinline__ uint32_t blend_f32(uint32_t a, uint32_t b, uint32_t c, uint32_t d, float wx, float wy)
{
    uint32_t r = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float pa = (float)((a >> (8 * k)) & 0xFFu), pb = (float)((b >> (8 * k)) & 0xFFu);
        const float pc = (float)((c >> (8 * k)) & 0xFFu), pd = (float)((d >> (8 * k)) & 0xFFu);
        const float h0 = __builtin_fmaf(wx, pb - pa, pa);
        const float h1 = __builtin_fmaf(wx, pd - pc, pc);
        r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(wy, h1 - h0, h0), k, r);
    }
    return r;
}

// The float16 form.  A byte needs no conversion at all to be a half: the bit pattern 0x00vv IS the (subnormal)
// float16 v * 2^-24.  v_perm_b32 puts the two VERTICAL neighbours of a channel into the two halves of a register,
// v_dot2_f32_f16 blends them with the pixel's float16 row weights ((1 - wy) 2^15, wy 2^15) and accumulates in float32
// (exact products: checked on the hardware, tools/ubench/cvt_probe.hip); the horizontal blend of the two column
// results is two float32 operations with the column weights pre-scaled by 2^9.  Per channel: 2 perm, 2 dot2, 2 float
// ops, 1 convert -- what the exact path's packed-integer blend costs.  Rows first (round 6): the row weights belong to
// the pixel, not to the yaw -- ONE packed register per pixel for the whole tile instead of two floats, and nothing to
// convert to float16 per yaw (the column weights, which change with every yaw's fractional shift, stay float32).
// A pixel with no footprint has both row weights 0 and comes out black.
// One pixel's six vertical blends, a.x * w.x + a.y * w.y in float32 each.  Written out as v_dot2_f32_f16 with the constant
// 0 as its addend: from the builtin the compiler makes the two-source v_dot2c_f32_f16, whose addend is its destination,
// and a v_mov_b32 0 in front of every one of them (48 of the loop's 306 vector instructions: config 5's f16 launch 871
// -> 790 us on one box).  The compiler does not know what an asm statement's instructions are, so the hazard it would
// have covered is covered here: on gfx940 and later a dot instruction's result may be read by another kind of vector
// instruction three wait states later at the earliest (LLVM's GCNHazardRecognizer: DotWriteDifferentVALURead) -- the
// s_nop behind the last of the six; the five in front of it are further than that from whatever follows the statement.
__device__ __forceinline__ void dot2_f16_x6(const f16x2 (&p)[6], f16x2 w, float (&v)[6])
{
#ifdef P2P_FLOAT_DOT2_BUILTIN
#pragma unroll
    for (int k = 0; k < 6; ++k)
        v[k] = __builtin_amdgcn_fdot2(p[k], w, 0.0f, false);
#else
    asm("v_dot2_f32_f16 %0, %6, %12, 0\n\t"
        "v_dot2_f32_f16 %1, %7, %12, 0\n\t"
        "v_dot2_f32_f16 %2, %8, %12, 0\n\t"
        "v_dot2_f32_f16 %3, %9, %12, 0\n\t"
        "v_dot2_f32_f16 %4, %10, %12, 0\n\t"
        "v_dot2_f32_f16 %5, %11, %12, 0\n\t"
        "s_nop 2"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(w));
#endif
}

__device__ __forceinline__ uint32_t blend_f16(uint32_t a, uint32_t b, uint32_t c, uint32_t d, f16x2 wy2, float wx0s, float wx1s)
{
    f16x2 p[6];
    float v[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t sel = 0x0C040C00u + 0x00010001u * (uint32_t)k;  // byte k of two pixels as two u16
        p[2 * k] = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(c, a, sel));      // left column: upper, lower
        p[2 * k + 1] = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(d, b, sel));  // right column
    }
    dot2_f16_x6(p, wy2, v);
    uint32_t r = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[2 * k + 1], wx1s, v[2 * k] * wx0s), k, r);
    return r;
}

// (1 - wy) 2^15 | wy 2^15 as two float16 (round to zero, as v_cvt_pkrtz_f16_f32 does); 0 | 0 for a pixel with no footprint
__device__ __forceinline__ f16x2 row_weights_f16(float wy, bool live)
{
    const float w1 = live ? wy * 32768.0f : 0.0f, w0 = live ? 32768.0f - wy * 32768.0f : 0.0f;
    return __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(w0, w1));
}

template <bool HALF>
__device__ __forceinline__ void draw_float(
    const ViewsParams& P, const uint8_t* __restrict__ src, uint8_t* __restrict__ out, const TileGeo& G,
    const uint32_t* __restrict__ pxw, const uint32_t* __restrict__ px2w, const uint32_t* __restrict__ itw,
    uint4 (*tile4)[LDS_ITEMS_CAP], uint32_t* stage)
{
    constexpr int PXT = VIEWS_PXT;
    const int t = threadIdx.x;
    if (G.mode != 1)
        return;  // the rest kernel's
    P2P_AUD_LT(P.audit, AUD_FLOAT_HDR, G.n_items, LDS_ITEMS_CAP + 1);
    const int pair0 = (P.chunk_outer ? blockIdx.z : blockIdx.y) * P.pairs_per_block;  // grid order as the exact kernel's
    int pair1 = pair0 + P.pairs_per_block;
    if (pair1 > P.n_panos * P.n_yaw)
        pair1 = P.n_panos * P.n_yaw;
    const int npairs = pair1 - pair0;
    if (npairs <= 0)
        return;

    // ---- this thread's pixels ----
    uint32_t tap_up[PXT], tap_lo[PXT];
    float fu[PXT], fus[PXT];  // the column fraction, and (float16 path) 512 times it
    float wyf[PXT];          // float32 path: the row weight
    f16x2 wy2[PXT];          // float16 path: both row weights, packed, times 2^15 (the column weights carry 2^9)
    bool live[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const uint32_t wd = pxw[j * VIEWS_BLOCK + t], fr = px2w[j * VIEWS_BLOCK + t];
        const uint32_t dl = (wd >> PXW_UP_BITS) & ((1u << PXW_DL_BITS) - 1u);
        tap_up[j] = (wd & ((1u << PXW_UP_BITS) - 1u)) << 2;
        tap_lo[j] = tap_up[j] + (dl << 2);
        live[j] = dl != 0u;
        fu[j] = (float)(fr & 0xFFFFu) * (1.0f / 65536.0f);
        fus[j] = fu[j] * 512.0f;
        const float wy = (float)(fr >> 16) * (1.0f / 65536.0f);
        wyf[j] = wy;
        wy2[j] = row_weights_f16(wy, live[j]);
    }
    uint32_t slot_off[VIEWS_SLOTS], slot_g[VIEWS_SLOTS];
    decode_items(itw, t, G.n_items, P.src_pitch, slot_off, slot_g);
    const uint32_t row_bytes = 3u * (uint32_t)P.pw;
    const size_t view_bytes = P.view_bytes;
    const int ngroups = P.pw >> 2;

    // ---- per-pair contexts in lane k: the yaw's integer column shift places the source pieces, its fraction is
    // added to every pixel's own ----
    uint32_t cw0 = 0, cw1 = 0;
    float cwf = 0.0f;
    int cw3 = 0;
    bool wanted = false;  // the job draws this pair's view of this pitch (p2p_job_set_view_mask)
    {
        const int k = t & 63;
        if (k < npairs) {
            cw3 = pano_of_pair(P, pair0 + k);
            const int yi = pair0 + k - cw3 * P.n_yaw;
            wanted = view_wanted(P, G.pitch_i, yi);
            const double sh = yaw_shift(P.yaw_rad[yi], P.pw);
            int si = (int)sh;
            if (si >= P.pw)
                si = P.pw - 1;
            cwf = (float)(sh - (double)si);
            int i_first = G.c0 + si;
            if (i_first >= P.pw)
                i_first -= P.pw;
            const int g0 = i_first >> 2;
            cw0 = 12u * (uint32_t)g0 | (uint32_t)(i_first & 3) << 20;
            cw1 = (uint32_t)(ngroups - g0);
        }
    }

    // the way out, as in the exact kernel: pixels -> LDS -> 4 adjacent pixels of one row per lane -> 12 bytes
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6), ln = t & 63;  // (the wave's index as a scalar: the staging address stays out of the VGPRs)
    uint32_t* const stg = stage + wv * (PXT * 64);
    const int x4 = 4 * (ln & 15), sj = ln >> 4;
    const int srow = ((wv * 64 + x4) >> TILE_LW) + sj * TILE_ROWSTEP, scol = (wv * 64 + x4) & (TILE_W - 1);
    const bool s_ok = sj < PXT && srow < TILE_H && G.y0 + srow < P.oh && G.x0 + scol < P.ow;
    const uint32_t out_off12 = s_ok ? (uint32_t)(G.y0 + srow) * (uint32_t)P.out_row + 3u * (uint32_t)(G.x0 + scol) : 0xFFFFFFFFu;
    const uint32_t stg_rd = (uint32_t)(sj * 64 + x4);

    const unsigned long long wanted_mask = __ballot(wanted);  // bit k: pair k of the chunk
    const int wave_base = __builtin_amdgcn_readfirstlane(t & ~63);
    int ns_wave = 0;
#pragma unroll
    for (int k = 0; k < VIEWS_SLOTS; ++k)
        ns_wave += G.n_items > wave_base + k * VIEWS_BLOCK;

    uint32_t buf_bytes = 0u;
    Q16 qc[VIEWS_SLOTS], qn[VIEWS_SLOTS];
    auto load_pieces = [&](auto ns_c, int k, Q16 (&qq)[VIEWS_SLOTS]) {
        constexpr int NS = decltype(ns_c)::value;
        const uint32_t goff = (uint32_t)__builtin_amdgcn_readlane((int)cw0, k) & 0xFFFFFu;
        const uint32_t wrap_g = (uint32_t)__builtin_amdgcn_readlane((int)cw1, k);
        const auto S = make_buf(src + (size_t)__builtin_amdgcn_readlane(cw3, k) * P.pano_stride, (uint32_t)P.pano_stride);
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            uint32_t off = slot_off[sl] + goff;
            off = slot_g[sl] >= wrap_g ? off - row_bytes : off;  // wrap-around: past the row's end, its start
            P2P_AUD_RANGE(P.audit, AUD_FLOAT_SRC, off, 16u, P.pano_stride);
            const bu32x4 q = __builtin_amdgcn_raw_buffer_load_b128(S, (int)off, 0, 0);
            qq[sl].d[0] = q.x; qq[sl].d[1] = q.y; qq[sl].d[2] = q.z; qq[sl].d[3] = q.w;
        }
    };
    auto run_ns = [&](auto ns_c) {
        constexpr int NS = decltype(ns_c)::value;
        load_pieces(ns_c, 0, qc);
        // one store that writes nothing, so that the loop is entered with the vmcnt shape it has inside
        // (see draw_tight in p2p_views.hip)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_buffer_store_b32(0u, __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000), 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        uint8_t* pend_O = out;
        uint32_t pend_records = 0u;
        auto flush_pending = [&]() {
            const uint4 v4 = *reinterpret_cast<const uint4*>(stg + stg_rd);
            u32x3 o;
            o.x = __builtin_amdgcn_perm(v4.y, v4.x, 0x04020100u);
            o.y = __builtin_amdgcn_perm(v4.z, v4.y, 0x05040201u);
            o.z = __builtin_amdgcn_perm(v4.w, v4.z, 0x06050402u);
            __builtin_amdgcn_raw_buffer_store_b96(o, __builtin_amdgcn_make_buffer_rsrc(pend_O, 0, (int)pend_records, 0x00020000),
                                                  (int)out_off12, 0, P2P_STORE_AUX);
        };
        // one pair: `cur` holds its source pieces, the next pair's are requested into `nxt` (the two sets swap roles from pair
        // to pair: no register copies)
        auto one_pair = [&](int k, const Q16 (&cur)[VIEWS_SLOTS], Q16 (&nxt)[VIEWS_SLOTS]) {
            uint4* tl4 = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(&tile4[0][0]) + buf_bytes);
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                // source pixels 0..3 of the piece (byte offsets 0, 3, 6, 9) as dwords
                const uint32_t d0 = cur[sl].d[0], d1 = cur[sl].d[1], d2 = cur[sl].d[2], d3 = cur[sl].d[3];
                uint4 o;
                o.x = d0 & 0x00FFFFFFu;
                o.y = __builtin_amdgcn_perm(d1, d0, 0x0C050403u);
                o.z = __builtin_amdgcn_perm(d2, d1, 0x0C040302u);
                o.w = __builtin_amdgcn_perm(d3, d2, 0x0C030201u);
                tl4[t + sl * VIEWS_BLOCK] = o;
            }
            uint32_t soff = buf_bytes + 4u * (((uint32_t)__builtin_amdgcn_readlane((int)cw0, k) >> 20) & 3u);
            asm volatile("" : "+s"(soff));
            const float sf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cwf), k));
            flush_pending();  // the pair before (the first pair: a descriptor of no records)
            __syncthreads();
            const unsigned char* tl = reinterpret_cast<const unsigned char*>(&tile4[0][0]);
            uint32_t ta[PXT][4];
            float wx[PXT];
            // (float16 path: fractions and weights times 512 = 2^9 throughout -- the scale the dot products' results are
            // short of; a power of two, so the sums and differences are the unscaled ones' times 512 exactly -- and the two tap
            // advances of a pair, with and without the carry, ready in registers: one select per pixel instead of select + add)
            const float sfs = HALF ? sf * 512.0f : sf;
            const float one = HALF ? 512.0f : 1.0f;
            uint32_t soff4 = soff + 4u;
            asm volatile("" : "+v"(soff4));
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                // fraction of the pixel's own coordinate + fraction of the yaw shift; a carry moves the taps one column on
                // (the sum with one added, v_fract_f32 for the weight and bit 30 of the pattern for the carry: ten vector
                // instructions fewer per iteration and not a microsecond faster -- docs/experiments.md)
                float xs = (HALF ? fus[j] : fu[j]) + sfs;
                const bool carry = xs >= one;
                wx[j] = carry ? xs - one : xs;
                const uint32_t adv = HALF ? (carry ? soff4 : soff) : soff + (carry ? 4u : 0u);
                const uint32_t* up = reinterpret_cast<const uint32_t*>(tl + (tap_up[j] + adv));
                const uint32_t* lo = reinterpret_cast<const uint32_t*>(tl + (tap_lo[j] + adv));
                ta[j][0] = up[0];
                ta[j][1] = up[1];
                ta[j][2] = lo[0];
                ta[j][3] = lo[1];
            }
            const int kn = k + 1 < npairs ? k + 1 : k;
            load_pieces(ns_c, kn, nxt);
            uint32_t pix[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                if (HALF) {
                    const float w1 = wx[j];  // (times 512 already)
                    pix[j] = blend_f16(ta[j][0], ta[j][1], ta[j][2], ta[j][3], wy2[j], 512.0f - w1, w1);
                    // (one pixel after the other: left to itself the scheduler interleaves the four blends, and the
                    // kernel's 80 registers -- six waves per SIMD -- no longer hold them)
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    const uint32_t v = blend_f32(ta[j][0], ta[j][1], ta[j][2], ta[j][3], wx[j], wyf[j]);
                    pix[j] = live[j] ? v : 0u;
                    if (TILE_W == 128)  // (the 128-wide instance spills without it; the 64-wide one is 2 % faster left alone)
                        __builtin_amdgcn_sched_barrier(0);
                }
            }
            // The way out runs ONE PAIR BEHIND (as the exact kernel's, p2p_views.hip): this pair's pixels go into the wave's
            // staging dwords and stay there; the NEXT pair reads them back, packs and stores them between its stage 1 and its
            // barrier -- where the wave waits for its LDS writes anyway -- so the staging round trip no longer stands between
            // the blend and the next pair.  (DS operations of one wave execute in order; the staging dwords are the wave's own.)
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                stg[j * 64 + ln] = pix[j];
            pend_O = out + ((size_t)(pair0 + k) * P.n_pitch + G.pitch_i) * view_bytes;  // [pano][yaw][pitch][oh][ow][3]
            // a view the job does not draw is stored through a descriptor of no records: dropped by the hardware
            pend_records = ((wanted_mask >> k) & 1ull) ? (uint32_t)view_bytes : 0u;
            buf_bytes ^= (uint32_t)sizeof(tile4[0]);
        };
        int k = 0;
        for (; k + 1 < npairs; k += 2) {
            one_pair(k, qc, qn);
            one_pair(k + 1, qn, qc);
        }
        if (k < npairs)
            one_pair(k, qc, qn);
        flush_pending();  // the last pair's pixels
    };
    static_assert(VIEWS_SLOTS >= 2 && VIEWS_SLOTS <= 4, "dispatch below");
    if (ns_wave == 0)
        run_ns(std::integral_constant<int, 0>{});
    else if (ns_wave == 1)
        run_ns(std::integral_constant<int, 1>{});
    else if (VIEWS_SLOTS == 2 || ns_wave == 2)
        run_ns(std::integral_constant<int, 2>{});
    else if (VIEWS_SLOTS == 3 || ns_wave == 3)
        run_ns(std::integral_constant<int, (VIEWS_SLOTS < 3 ? VIEWS_SLOTS : 3)>{});
    else
        run_ns(std::integral_constant<int, VIEWS_SLOTS>{});
}

template <bool HALF>
#ifndef P2P_FLOAT_HALF_WAVES
#define P2P_FLOAT_HALF_WAVES 6  // (f16: 77 VGPRs, no scratch; at 5 waves per SIMD config 5 runs 1 % faster -- 837 against 847 us -- and keeps fewer workgroups per CU)
#endif
__global__ __launch_bounds__(VIEWS_BLOCK, HALF ? P2P_FLOAT_HALF_WAVES : 6) void float_views_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, uint8_t* __restrict__ out,
    const PieceHdr* __restrict__ hdr, const uint32_t* __restrict__ px, const uint32_t* __restrict__ px2,
    const uint32_t* __restrict__ items)
{
    __shared__ uint4 tile4[2][LDS_ITEMS_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t stage[(VIEWS_BLOCK / 64) * VIEWS_PXT * 64];
    const int tile_id = tile_of_block(P, (int)blockIdx.x, (int)gridDim.x);
    if (tile_id < 0)
        return;
    const int pitch_i = pitch_of_block(P, (int)(P.chunk_outer ? blockIdx.y : blockIdx.z));
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const PieceHdr h = hdr[(size_t)pitch_i * tiles + tile_id];
    const TileGeo G = tile_geo(P, h, pitch_i, tile_id, (int)threadIdx.x);
    const size_t pb = (size_t)G.slot * (VIEWS_BLOCK * VIEWS_PXT);
    draw_float<HALF>(P, src, out, G, px + pb, px2 + pb, items + (size_t)G.slot * LDS_ITEMS_CAP, tile4, stage);
}

// Tiles the plan marks for gathers: one thread per pixel, taps from global memory (the view's own seam,
// where U jumps from pw - 1 to 0 between neighbouring pixels; a pole inside the tile; widths not divisible by 4).
template <bool HALF>
__global__ __launch_bounds__(VIEWS_BLOCK) void float_views_rest_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, uint8_t* __restrict__ out, const PieceHdr* __restrict__ hdr)
{
    // one workgroup per (tile of the plan's gather list, chunk of pairs)
    const uint32_t tiles = (uint32_t)(((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H));
    uint32_t slot = P.gather_list[blockIdx.x];
    P2P_AUD_LT(P.audit, AUD_FLOAT_LIST, slot, tiles * (uint32_t)P.n_pitch);
    if (slot >= tiles * (uint32_t)P.n_pitch)
        slot = 0u;
    const int pitch_i = (int)(slot / tiles), tile_id = (int)(slot - (uint32_t)pitch_i * tiles);
    const PieceHdr h = hdr[slot];
    const int t = threadIdx.x;
    const TileGeo G = tile_geo(P, h, pitch_i, tile_id, t);
    if (G.mode != 2)
        return;
    const int pair0 = blockIdx.z * P.gather_ppb;
    int pair1 = pair0 + P.gather_ppb;
    if (pair1 > P.n_panos * P.n_yaw)
        pair1 = P.n_panos * P.n_yaw;
    const size_t view_bytes = P.view_bytes;
    const int px = G.x0 + G.col;
    for (int j = 0; j < VIEWS_PXT; ++j) {
        const int row = G.row0 + j * TILE_ROWSTEP, py = G.y0 + row;
        if (row >= TILE_H || px >= P.ow || py >= P.oh)
            continue;
        const int2 cc = P.coords[((size_t)G.pitch_i * P.oh + py) * P.ow + px];
        const float U = __int_as_float(cc.x), V = __int_as_float(cc.y);
        const bool dead = !(U == U);
        int y0 = dead ? 0 : (int)V;  // V in [0, ph - 1]
        y0 = y0 < 0 ? 0 : (y0 > P.ph - 1 ? P.ph - 1 : y0);
        const float wy = dead ? 0.0f : V - (float)y0;
        const int y1 = y0 + 1 < P.ph ? y0 + 1 : y0;
        const size_t px_off = (size_t)py * P.out_row + 3 * (size_t)px;
        for (int pair = pair0; pair < pair1; ++pair) {
            const int pano = pano_of_pair(P, pair), yi = pair - pano * P.n_yaw;
            if (!view_wanted(P, G.pitch_i, yi))
                continue;
            uint8_t* O = out + ((size_t)pair * P.n_pitch + G.pitch_i) * view_bytes + px_off;
            if (dead) {
                O[0] = O[1] = O[2] = 0;
                continue;
            }
            const uint8_t* __restrict__ S = src + (size_t)pano * P.pano_stride;
            const uint8_t* __restrict__ r0 = S + (size_t)y0 * P.src_pitch;
            const uint8_t* __restrict__ r1 = S + (size_t)y1 * P.src_pitch;
            float xs = U + (float)yaw_shift(P.yaw_rad[yi], P.pw);
            if (xs >= (float)P.pw)
                xs -= (float)P.pw;
            int x0 = (int)xs;
            x0 = x0 < 0 ? 0 : (x0 >= P.pw ? P.pw - 1 : x0);
            const float wx = xs - (float)x0;
            const int x1 = x0 + 1 < P.pw ? x0 + 1 : 0;  // wrap-around
            uint32_t a = 0, b = 0, c = 0, d = 0;
            __builtin_memcpy(&a, r0 + 3 * x0, 3);
            __builtin_memcpy(&b, r0 + 3 * x1, 3);
            __builtin_memcpy(&c, r1 + 3 * x0, 3);
            __builtin_memcpy(&d, r1 + 3 * x1, 3);
            uint32_t res;
            if (HALF) {
                const float w1 = wx * 512.0f;
                res = blend_f16(a, b, c, d, row_weights_f16(wy, true), 512.0f - w1, w1);
            } else {
                res = blend_f32(a, b, c, d, wx, wy);
            }
            O[0] = (uint8_t)res;
            O[1] = (uint8_t)(res >> 8);
            O[2] = (uint8_t)(res >> 16);
        }
    }
}

// which = 0: the tile kernel, 1: the gather tiles
hipError_t launch_float_views(const ViewsParams& P, bool half, int which, hipStream_t st)
{
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const int n_pairs = P.n_panos * P.n_yaw;
    const int zblocks = (n_pairs + P.pairs_per_block - 1) / P.pairs_per_block;
    const dim3 grid(8 * ((tiles + 7) / 8), P.chunk_outer ? P.n_pitch : zblocks, P.chunk_outer ? zblocks : P.n_pitch);
    if (which == 0) {
        if (half)
            hipLaunchKernelGGL(float_views_kernel<true>, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.out, P.hdr, P.px, P.px2, P.items);
        else
            hipLaunchKernelGGL(float_views_kernel<false>, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.out, P.hdr, P.px, P.px2, P.items);
    } else {
        const dim3 dgrid(P.n_gather, 1, (n_pairs + P.gather_ppb - 1) / P.gather_ppb);
        if (half)
            hipLaunchKernelGGL(float_views_rest_kernel<true>, dgrid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.out, P.hdr);
        else
            hipLaunchKernelGGL(float_views_rest_kernel<false>, dgrid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.out, P.hdr);
    }
    return hipGetLastError();
}

}  // namespace P2P_SHAPE_NS
}  // namespace p2p
