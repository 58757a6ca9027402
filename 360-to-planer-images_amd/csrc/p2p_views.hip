// p2p_views.hip -- the hot kernel: both cv2.remap stages of every (panorama, yaw, pitch) view in one launch
//   cv2.remap x2       P:192-199, P:212-218 -> remap_views_kernel (both stages fused, fixed point),
//                                            remap_views_rest_kernel and remap_views_direct_kernel (the odd
//                                            cases), driven by the tables of the plan pass (p2p_plan.hip);
//                                            3-channel single remaps of the legacy tool (L:179) too
// Reference behaviour (cited, never copied):
//   P = /root/reference/app/panorama_to_plane-pitch.py, L = /root/reference/app/legacy/panorama_to_plane.py
// The fixed-point arithmetic is OpenCV 4.10's (imgwarp.cpp remapBilinear, INTER_BITS = 5,
// INTER_REMAP_COEF_BITS = 15); see DESIGN.md "Arithmetic contract".
// Compiled with -ffp-contract=off: every float operation below rounds where NumPy rounds.
#include <type_traits>
#include "p2p_tile.h"

namespace p2p {

// ---------------------------------------------------------------------------------------------
// Stage 1, P:192-199: one pixel of the yaw-resampled panorama ("rot") from two horizontally
// adjacent source pixels p0, p1 (dwords B | G<<8 | R<<16 | x<<24):
//   rot = ((32-f)*p0 + f*p1 + 16) >> 5 per channel
// which is cv::remap with fy == 0 (weights 1024*(32-f), 1024*f, rounding 1<<14, shift 15).
// Two channels share one 32-bit multiply (16-bit fields hold <= 32*255 + 16).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rot_blend2(uint32_t p0, uint32_t p1, uint32_t f, uint32_t g)
{
    uint32_t br = g * (p0 & 0x00FF00FFu) + f * (p1 & 0x00FF00FFu) + 0x00100010u;
    uint32_t gg = g * (p0 & 0x0000FF00u) + f * (p1 & 0x0000FF00u) + 0x00001000u;
    return ((br >> 5) & 0x00FF00FFu) | ((gg >> 5) & 0x0000FF00u);
}

// The same value with the weights pre-multiplied by 8 (f8 = 8f, g8 = 8g): every 16-bit field then holds
// 8*(g*a + f*b + 16) <= 65408, so the wanted byte (sum >> 5) is simply the field's HIGH byte and one
// v_perm_b32 assembles B | G<<8 | R<<16 -- no shifts, no masks on the way out.
//   m0 = p & 0x00FF00FF (B, R fields), m1 = p & 0x0000FF00 (G field) of the left / right source pixel.
__device__ __forceinline__ uint32_t umad24(uint32_t a, uint32_t b, uint32_t c)
{
    return (uint32_t)__umul24(a, b) + c;  // v_mad_u32_u24: both factors fit 24 bits
}

__device__ __forceinline__ uint32_t rot_blend8(uint32_t a_br, uint32_t a_g, uint32_t b_br, uint32_t b_g,
                                               uint32_t f8, uint32_t g8)
{
    const uint32_t br = umad24(f8, b_br, umad24(g8, a_br, 0x00800080u));  // bytes 1 and 3
    const uint32_t gg = umad24(f8, b_g, umad24(g8, a_g, 0x00008000u));    // byte 2
    // v_perm_b32(S0, S1, sel): selector bytes 0-3 pick S1's bytes, 4-7 pick S0's, 0x0c is zero
    return __builtin_amdgcn_perm(br, gg, 0x0C070205u);
}

// The tight loops' form: v_mad_u32_u24 twice per field pair.  Left to itself the compiler shares the products
// between neighbouring pixels and adds them with v_add3_u32 (24 instructions per 4-pixel item instead of 16),
// because a mad with the scalar weight AND a literal rounding constant would need two constant-bus reads: the
// constants are therefore handed in as VGPRs.
__device__ __forceinline__ uint32_t vmad24(uint32_t s_w, uint32_t v, uint32_t c)
{
#ifdef P2P_NO_ASM_MAD
    return umad24(s_w, v, c);
#else
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "s"(s_w), "v"(v), "v"(c));
    return r;
#endif
}

// direct path: (3*i | f << 20) table entry, unaligned 8-byte load of pixels i and i+1
__device__ __forceinline__ uint32_t rot_pixel(const uint8_t* __restrict__ row, uint32_t te)
{
    uint2 q;
    __builtin_memcpy(&q, row + (te & 0xFFFFFu), 8);
    const uint32_t f = te >> 20;
    return rot_blend2(q.x, __builtin_amdgcn_alignbyte(q.y, q.x, 3), f, 32u - f);
}

// Stage 2 for one output pixel, P:212-218: bilinear blend of four rot pixels with cv::remap's
// weights 32*(32-fx)(32-fy).. and (sum + 16384) >> 15  ==  (sum' + 512) >> 10 with weights / 32.
__device__ __forceinline__ uint32_t blend4(uint32_t a, uint32_t b, uint32_t c, uint32_t d,
                                           uint32_t fx, uint32_t fy)
{
    uint32_t gx = 32u - fx, gy = 32u - fy;
    // horizontal: two channels per 32-bit op (16-bit fields hold <= 32*255)
    uint32_t h0br = gx * (a & 0x00FF00FFu) + fx * (b & 0x00FF00FFu);
    uint32_t h1br = gx * (c & 0x00FF00FFu) + fx * (d & 0x00FF00FFu);
    uint32_t h0g = gx * ((a >> 8) & 0xFFu) + fx * ((b >> 8) & 0xFFu);
    uint32_t h1g = gx * ((c >> 8) & 0xFFu) + fx * ((d >> 8) & 0xFFu);
    uint32_t vb = (gy * (h0br & 0xFFFFu) + fy * (h1br & 0xFFFFu) + 512u) >> 10;
    uint32_t vr = (gy * (h0br >> 16) + fy * (h1br >> 16) + 512u) >> 10;
    uint32_t vg = (gy * h0g + fy * h1g + 512u) >> 10;
    return vb | (vg << 8) | (vr << 16);
}


// per-pixel stage-2 weights, constant across (panorama, yaw) pairs: cv::remap's four tap weights
// (32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx*fy (sum 1024), scaled by 64 so that the rounded result
// (sum + 512) >> 10 is byte 2 of the scaled sum.  The one weight that does not fit 16 bits, 1024 * 64 (fx = fy = 0:
// the other three are 0), is stored as 65504 (tap_weights below): byte 2 comes out the same.
struct TapWeights {
    uint32_t w_up;  // [w_a, w_b] as two u16 (0 for a pixel with no footprint)
    uint32_t w_lo;  // [w_c, w_d]
};

__device__ __forceinline__ TapWeights tap_weights(uint32_t fx, uint32_t fy, bool live)
{
    // [32-fx, fx] as two u16 times 64 (32-fy) and times 64 fy, one packed multiply each; for fx = fy = 0 the row
    // factor is 2047 instead of 2048: a * 32 * 2047 + 32768 = a * 65536 + (32768 - 32 a) keeps byte 2 = a, where
    // 32 * 2048 would not fit the u16
    const u16x2 wx = as_u16x2((32u - fx) | (fx << 16));
    const uint32_t up = 2048u - 64u * fy - ((fx | fy) == 0u ? 1u : 0u), lo = 64u * fy;
    TapWeights w;
    w.w_up = live ? __builtin_bit_cast(uint32_t, wx * as_u16x2(up | (up << 16))) : 0u;
    w.w_lo = live ? __builtin_bit_cast(uint32_t, wx * as_u16x2(lo | (lo << 16))) : 0u;
    return w;
}

// Stage 2 as two 2-tap dot products per channel: v_perm_b32 widens one channel of two horizontal neighbours to two
// u16, v_dot2_u32_u16 multiplies by the row's two weights and accumulates (exact in 32 bits: <= 255 * 65536 + 32768).
// 14 instructions per pixel; identical in value to blend4().
__device__ __forceinline__ uint32_t blend4_packed(uint32_t a, uint32_t b, uint32_t c, uint32_t d,
                                                  const TapWeights& w)
{
    const u16x2 wu = as_u16x2(w.w_up), wl = as_u16x2(w.w_lo);
    // v_perm_b32(S0, S1, sel): selector bytes 0-3 pick S1's bytes, 4-7 pick S0's, 0x0c is zero
    const u16x2 b_ab = as_u16x2(__builtin_amdgcn_perm(b, a, 0x0C040C00u));
    const u16x2 b_cd = as_u16x2(__builtin_amdgcn_perm(d, c, 0x0C040C00u));
    const u16x2 g_ab = as_u16x2(__builtin_amdgcn_perm(b, a, 0x0C050C01u));
    const u16x2 g_cd = as_u16x2(__builtin_amdgcn_perm(d, c, 0x0C050C01u));
    const u16x2 r_ab = as_u16x2(__builtin_amdgcn_perm(b, a, 0x0C060C02u));
    const u16x2 r_cd = as_u16x2(__builtin_amdgcn_perm(d, c, 0x0C060C02u));
    const uint32_t vb = __builtin_amdgcn_udot2(b_cd, wl, __builtin_amdgcn_udot2(b_ab, wu, 32768u, false), false);
    const uint32_t vg = __builtin_amdgcn_udot2(g_cd, wl, __builtin_amdgcn_udot2(g_ab, wu, 32768u, false), false);
    const uint32_t vr = __builtin_amdgcn_udot2(r_cd, wl, __builtin_amdgcn_udot2(r_ab, wu, 32768u, false), false);
    const uint32_t bg = __builtin_amdgcn_perm(vg, vb, 0x0C0C0602u);  // B | G << 8
    return __builtin_amdgcn_perm(vr, bg, 0x0C060100u);               // | R << 16
}

// ---------------------------------------------------------------------------------------------
// The view kernels.  One workgroup = one piece of the plan (p2p_plan.hip): a tile of TILE_W x TILE_H output
// pixels of one pitch view (VIEWS_PXT pixels per thread), or a part of a tile whose footprint did not fit.  The
// plan pass has already worked out everything that depends on the maps only, so a workgroup starts with a handful
// of loads: the piece header (scalar), one dword per pixel (LDS offsets of its 2x2 taps + the two 5-bit weights)
// and one dword per footprint item (rot row, 4-pixel group).  Then, per (panorama, yaw) pair of its chunk:
//   stage 1  the yaw map is a circular column shift (YawDesc), so a footprint row is one contiguous run of
//            source bytes: each thread loads one 4-byte-aligned 16-byte piece (5 1/3 source pixels: fully
//            coalesced, no per-pixel table lookup), blends 4 rot pixels in registers with the exact uint8
//            arithmetic and writes them to the LDS tile (double-buffered) with one ds_write_b128; the loads of
//            the NEXT pair are issued before stage 2 so that their latency hides behind it;
//   stage 2  after one barrier each thread reads the 2x2 taps of its pixels from LDS, blends with cv::remap's
//            fixed-point weights; the pixels of a wave go through LDS once more so that every lane ends up with
//            4 adjacent pixels of one row = one 12-byte non-temporal store.
// The footprint is kept as per-row spans (each rot row only as wide as the taps of that row need), not as the
// bounding rectangle: 1.4 .. 1.7 rot pixels per output pixel instead of 1.7 .. 2.3 on config 2.
//
// Three kernels share this arithmetic:
//   remap_views_kernel         whole interior pieces x yaws that are plain shifts -- nothing but three branch-free
//                              loops (copy / blend / blend with the clipped column patched), 78 VGPRs, no spills;
//   remap_views_rest_kernel    the general LDS loop with its case distinctions: yaws with per-column weights, yaw
//                              rows that are not a shift (gathered per pixel), byte stores for view widths not
//                              divisible by 4.  With view rows of whole dwords only the pairs of the job's
//                              rest_pairs list, several per workgroup;
//   remap_views_direct_kernel  the pieces the plan marks for direct gathers (a pole inside the piece: the footprint
//                              spans every column; footprints touching the panorama border; general caller maps
//                              with border taps), one workgroup per (piece of the plan's list, chunk of pairs).
// Kept in one kernel, the rare paths set the register allocation (96 VGPRs + spills) and tripled the ISA; the
// common path ran at 110 us instead of 75 on config 2.  The kernels write disjoint pixels; the last two are
// launched (same stream, before the main one) only when the plan or the yaw tables have something for them.
// Blocks map to tiles XCD-aware: each of the 8 XCDs owns a contiguous run of the tile raster, so neighbouring
// tiles (shared source halo and output lines) meet in one L2.
// ---------------------------------------------------------------------------------------------
// the pieces the main kernel draws: LDS scheme, view rows of whole dwords (its stores are 12 bytes = 4 pixels)
template <int PXT>
__device__ __forceinline__ bool tight_piece(const PieceGeo& g, const ViewsParams& P)
{
    return g.mode == 1 && (P.ow & 3) == 0 && (PXT == 4 ? g.w == 64 : (g.w == 32 || g.w == 16));
}

// ---- per-pair contexts: lane k of every wave works out pair pair0 + k once; the loops read them back with
// v_readlane, so no descriptor load sits on the per-pair critical path.  Sorted by class across the lanes:
//   0  whole-column shift (stage 1 is a copy), footprint inside one pass of the source row     } tight loops
//   1  whole-column shift, footprint across the source row's end (items wrap to its start)      }
//   2  one blend weight for the whole piece, footprint inside one pass of the row               }
//   3  one blend weight, footprint across the row's end -- where the column P:105 clips to      }
//      pw - 1 lives: that one rot pixel is a copy instead
//   4  per-column weights (a shift fraction within float noise of a rounding tie: 6 of the 360 one-degree
//      yaws on 8192 columns) or a caller row that is not a shift -> general loop of the rest kernel
struct PairCtxs {
    uint32_t cw0, cw1;
    int cw2, cw3;
    int n0, n1, n2, n3;  // pairs of class 0, of classes 0..1, 0..2, 0..3
    int npairs, pair0;
};

// LIST: the chunk's pairs come from P.rest_pairs (the rest kernel drawing only the yaws left to it) instead of being
// the contiguous run blockIdx.z * pairs_per_block ...
template <bool LIST = false>
__device__ __forceinline__ PairCtxs pair_contexts(const ViewsParams& P, const YawDesc* __restrict__ ydesc, int c0, int c1, int t)
{
    PairCtxs X;
    const int ppb = LIST ? P.rest_ppb : P.pairs_per_block;
    X.pair0 = blockIdx.z * ppb;
    int pair1 = X.pair0 + ppb;
    const int n_pairs = LIST ? P.n_rest_pairs : P.n_panos * P.n_yaw;
    if (pair1 > n_pairs)
        pair1 = n_pairs;
    X.npairs = pair1 - X.pair0;
    const int ngroups = P.pw >> 2;
    uint32_t cw0 = 0, cw1 = 0;
    int cw2 = 0, cw3 = 0, cls = 4;
    const int k = t & 63;
    if (k < X.npairs) {
        const int pair = LIST ? (int)P.rest_pairs[X.pair0 + k] : X.pair0 + k;
        cw3 = pano_of_pair(P, pair);
        const int yi = pair - cw3 * P.n_yaw;
        const YawDesc yd = ydesc[yi];
        int i_first = c0 + yd.s;
        if (i_first >= P.pw)
            i_first -= P.pw;
        const int g0 = i_first >> 2;
        const bool clamp_in = yd.c_clamp >= c0 && yd.c_clamp <= c1 + 1;
        // uniform weight unless this yaw flickers or the piece holds the column clipped to pw-1
        const bool per_column = yd.mode == 1 || clamp_in;
        cw0 = 12u * (uint32_t)g0 | (uint32_t)(i_first & 3) << 20 | (uint32_t)(yd.mode != 2) << 22 |
              (uint32_t)per_column << 23 | (uint32_t)yd.f << 24;
        cw1 = (uint32_t)(ngroups - g0) | (uint32_t)yi << 16;
        cw2 = 4 * g0 - yd.s;
        cw3 |= k << 26;  // n_panos < 2^26 (host check): the chunk-local pair index rides along
        // items reach group ((c1 + 1 - c0) + 3) >> 2 past the first; beyond the row's last group they wrap
        // (telling the two apart at compile time would save 3 instructions per item, but the prefetch of the NEXT
        // pair crosses class boundaries: not done, only the clipped column makes a class of its own)
        const bool wraps = clamp_in;
        cls = yd.mode != 0 ? 4 : (yd.f == 0 ? (wraps ? 1 : 0) : (wraps ? 3 : 2));
    }
    const bool valid = k < X.npairs;
    const unsigned long long below = (1ull << k) - 1ull;
    int r = k, base = 0, cum[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const unsigned long long m = __ballot(valid && cls == c);
        if (valid && cls == c)
            r = base + __popcll(m & below);
        base += __popcll(m);
        cum[c] = base;
    }
    X.n0 = cum[0];
    X.n1 = cum[1];
    X.n2 = cum[2];
    X.n3 = cum[3];
    X.cw0 = (uint32_t)__builtin_amdgcn_ds_permute(4 * r, (int)cw0);
    X.cw1 = (uint32_t)__builtin_amdgcn_ds_permute(4 * r, (int)cw1);
    X.cw2 = __builtin_amdgcn_ds_permute(4 * r, cw2);
    X.cw3 = __builtin_amdgcn_ds_permute(4 * r, cw3);
    return X;
}

// per-pixel words of the plan -> tap offsets (bytes inside one LDS buffer) and packed weights
template <int PXT>
__device__ __forceinline__ void decode_px(const uint32_t* __restrict__ pxw, int t, uint32_t (&tap_up)[PXT],
                                          uint32_t (&tap_lo)[PXT], TapWeights (&tw)[PXT])
{
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const uint32_t wd = pxw[j * VIEWS_BLOCK + t];
        const uint32_t dl = (wd >> PXW_UP_BITS) & ((1u << PXW_DL_BITS) - 1u);
        tap_up[j] = (wd & ((1u << PXW_UP_BITS) - 1u)) << 2;
        tap_lo[j] = tap_up[j] + (dl << 2);
        const uint32_t fx = (wd >> 22) & 31u, fy = wd >> 27;
        // a pixel with no footprint in the panorama (NaN coordinate) gets weight 0 everywhere:
        // (0 + 512) >> 10 == 0, the BORDER_CONSTANT value
        tw[j] = tap_weights(fx, fy, dl != 0);
    }
}

// ---------------------------------------------------------------------------------------------
// Main kernel body: the three tight loops.  Free of branches on the vector-memory path, so that the compiler's
// s_waitcnt vmcnt stay counted (with a conditional load or store in the loop it falls back to vmcnt(0), and every
// pair then waits for the previous pair's stores to be acknowledged: loads and stores retire in issue order).
// ---------------------------------------------------------------------------------------------
template <int PXT>
__device__ __forceinline__ void draw_tight(
    const ViewsParams& P, const uint8_t* __restrict__ src, const YawDesc* __restrict__ ydesc,
    uint8_t* __restrict__ out, const PieceHdr h, const uint32_t* __restrict__ pxw, const uint32_t* __restrict__ itw,
    uint4 (*tile4)[LDS_ITEMS_CAP], uint32_t* stage)
{
    const int t = threadIdx.x;
    const PieceGeo G = piece_geo(h, t);
    if (!tight_piece<PXT>(G, P))
        return;  // the rest kernel's
    const PairCtxs X = pair_contexts(P, ydesc, h.c0, h.c1, t);
    const int nplain = X.n3;
    if (nplain == 0)
        return;
    uint32_t tap_up[PXT], tap_lo[PXT];
    TapWeights tw[PXT];
    decode_px<PXT>(pxw, t, tap_up, tap_lo, tw);
#ifdef P2P_ABLATE_CONFLICTS
    // timing experiment only (wrong pixels): every lane reads its own two dwords -- tap reads without bank conflicts
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        tap_up[j] = (uint32_t)(t & 63) * 8u + (uint32_t)j * 512u;
        tap_lo[j] = tap_up[j] + 2048u;
    }
#endif
    uint32_t slot_off[VIEWS_SLOTS], slot_g[VIEWS_SLOTS];
    decode_items(itw, t, G.n_items, P.src_pitch, slot_off, slot_g);
    const uint32_t row_bytes = 3u * (uint32_t)P.pw;
    const size_t view_bytes = (size_t)P.oh * P.ow * 3;

    // the way out: a wave's pixels -> LDS (a dword per pixel) -> 4 adjacent pixels of one row per lane -> 12 bytes,
    // written with a buffer store whose descriptor covers exactly this view: lanes with nothing to store (rows past
    // the piece or the view, four-pixel groups the piece does not have) get an offset beyond it and the hardware
    // drops them.  No lane is masked off: a masked store brings a branch, and with it vmcnt(0).
    const int wv = t >> 6, ln = t & 63;
#ifndef P2P_DPP_STORES
    uint32_t* const stg = stage + wv * (PXT * 64);
    const int x4 = 4 * (ln & 15), sj = ln >> 4;    // the group's first pixel as a lane of this wave; which of the thread's pixels
    const int srow = ((wv * 64 + x4) >> G.lw) + sj * G.rstep, scol = x4 & (G.w - 1);
    const bool s_ok = sj < PXT && srow < G.h && G.y0 + srow < P.oh && G.x0 + scol < P.ow;
    const uint32_t out_off12 = s_ok ? (uint32_t)(((size_t)(G.y0 + srow) * P.ow + G.x0 + scol) * 3) : 0xFFFFFFFFu;
    const uint32_t stg_rd = (uint32_t)(sj * 64 + x4);
#else
    // Four adjacent pixels of a row are 12 bytes = 3 dwords: lane 4m + r (r = 0, 1, 2) makes dword r of its group
    // from its own pixel and its right neighbour's (one DPP row shift, one v_perm_b32), lane 4m + 3 stores nothing.
    (void)stage; (void)wv;
    const int r4 = ln & 3;
    const uint32_t out_sel = r4 == 0 ? 0x04020100u : r4 == 1 ? 0x05040201u : 0x06050402u;
    uint32_t out_off[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const int row = G.row0 + j * G.rstep;
        const bool ok = r4 != 3 && row < G.h && G.y0 + row < P.oh && G.x0 + G.col < P.ow;
        out_off[j] = ok ? (uint32_t)(((size_t)(G.y0 + row) * P.ow + G.x0 + G.col) * 3) + (uint32_t)r4 : 0xFFFFFFFFu;
#ifdef P2P_ABLATE_STORES3
        out_off[j] = 0xFFFFFFFFu;  // timing experiment: every store issued, every store dropped by the range check
#endif
    }
#endif

    const int wave_base = __builtin_amdgcn_readfirstlane(t & ~63);
    int ns_wave = 0;  // items this wave produces per pair (wave-uniform)
#pragma unroll
    for (int k = 0; k < VIEWS_SLOTS; ++k)
        ns_wave += G.n_items > wave_base + k * VIEWS_BLOCK;

    uint32_t bias_br = 0x00800080u;  // rounding of both 16-bit fields, kept in a VGPR (see vmad24)
    asm volatile("" : "+v"(bias_br));

    uint32_t buf_bytes = 0u;
    Q16 qc[VIEWS_SLOTS], qn[VIEWS_SLOTS];
    auto load_pieces = [&](auto ns_c, int k, Q16 (&qq)[VIEWS_SLOTS]) {
        constexpr int NS = decltype(ns_c)::value;
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)X.cw0, k);
        const uint32_t wrap_g = (uint32_t)__builtin_amdgcn_readlane((int)X.cw1, k) & 0xFFFFu;
        const uint8_t* __restrict__ S = src + (size_t)(__builtin_amdgcn_readlane(X.cw3, k) & 0x3FFFFFF) * P.pano_stride;
        const uint32_t goff = w0 & 0xFFFFFu;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            uint32_t off = slot_off[sl] + goff;
            off = slot_g[sl] >= wrap_g ? off - row_bytes : off;  // items past the end of the row continue at its start
#ifdef P2P_ABLATE_LOADS2
            off &= 0x3FFFu;  // timing experiment (wrong pixels): every load issued, all of them hits in 16 KB
#endif
            qq[sl] = *reinterpret_cast<const Q16*>(S + off);
        }
    };
    // MODE 0: copy, 1: blend, 2: blend, and the rot pixel whose source column is pw - 1 (P:105's clip) is a copy
    auto stage1 = [&](auto ns_c, auto mode_c, int k, const Q16 (&qq)[VIEWS_SLOTS], uint4* tl4) {
        constexpr int NS = decltype(ns_c)::value;
        constexpr int MODE = decltype(mode_c)::value;
        const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)X.cw0, k) >> 24;
        const uint32_t last_g = ((uint32_t)__builtin_amdgcn_readlane((int)X.cw1, k) & 0xFFFFu) - 1u;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            // the piece holds source pixels 0..4 at byte offsets 0, 3, 6, 9, 12; one v_perm_b32
            // both fetches a pixel across the dword seam and masks it
            // (selector bytes 0-3 pick the second operand's bytes, 4-7 the first's, 0x0c is zero)
            const uint32_t d0 = qq[sl].d[0], d1 = qq[sl].d[1], d2 = qq[sl].d[2], d3 = qq[sl].d[3];
            uint4 o;
            if (MODE != 0) {
                const uint32_t f8 = 8u * f, g8 = 256u - f8;
                // B and R of a pixel share one multiply-add pair (fields at bits 0 and 16), and so do the G of two
                // neighbouring pixels: 12 v_mad_u32_u24 per item instead of 16.  Source byte 3i + c is channel c of
                // source pixel i; the piece's dwords hold bytes 0-3, 4-7, 8-11, 12-15.
                const uint32_t m0 = d0 & 0x00FF00FFu;                                // B0 R0  (bytes 0, 2)
                const uint32_t m1 = __builtin_amdgcn_perm(d1, d0, 0x0C050C03u);      // B1 R1  (3, 5)
                const uint32_t m2 = __builtin_amdgcn_perm(d2, d1, 0x0C040C02u);      // B2 R2  (6, 8)
                const uint32_t m3 = __builtin_amdgcn_perm(d3, d2, 0x0C030C01u);      // B3 R3  (9, 11)
                const uint32_t m4 = d3 & 0x00FF00FFu;                                // B4 R4  (12, 14)
                const uint32_t n01 = __builtin_amdgcn_perm(d1, d0, 0x0C040C01u);     // G0 G1  (1, 4)
                const uint32_t n12 = __builtin_amdgcn_perm(d1, d1, 0x0C030C00u);     // G1 G2  (4, 7)
                const uint32_t n23 = __builtin_amdgcn_perm(d2, d1, 0x0C060C03u);     // G2 G3  (7, 10)
                const uint32_t n34 = __builtin_amdgcn_perm(d3, d2, 0x0C050C02u);     // G3 G4  (10, 13)
                const uint32_t br0 = vmad24(f8, m1, vmad24(g8, m0, bias_br));
                const uint32_t br1 = vmad24(f8, m2, vmad24(g8, m1, bias_br));
                const uint32_t br2 = vmad24(f8, m3, vmad24(g8, m2, bias_br));
                const uint32_t br3 = vmad24(f8, m4, vmad24(g8, m3, bias_br));
                const uint32_t g01 = vmad24(f8, n12, vmad24(g8, n01, bias_br));
                const uint32_t g23 = vmad24(f8, n34, vmad24(g8, n23, bias_br));
                // every 16-bit field holds 256 * blend + rounding: the wanted byte is the field's high byte
                o.x = __builtin_amdgcn_perm(br0, g01, 0x0C070105u);
                o.y = __builtin_amdgcn_perm(br1, g01, 0x0C070305u);
                o.z = __builtin_amdgcn_perm(br2, g23, 0x0C070105u);
                o.w = __builtin_amdgcn_perm(br3, g23, 0x0C070305u);
                if (MODE == 2) {
                    // source column pw - 1 is pixel 3 of the row's last item; its right neighbour is not a pixel
                    // of this row, and the clipped map gives it weight 0 anyway
                    const uint32_t cp = __builtin_amdgcn_perm(d3, d2, 0x0C030201u);
                    o.w = slot_g[sl] == last_g ? cp : o.w;
                }
            } else {
                o.x = d0 & 0x00FFFFFFu;
                o.y = __builtin_amdgcn_perm(d1, d0, 0x0C050403u);
                o.z = __builtin_amdgcn_perm(d2, d1, 0x0C040302u);
                o.w = __builtin_amdgcn_perm(d3, d2, 0x0C030201u);
            }
            tl4[t + sl * VIEWS_BLOCK] = o;
        }
    };
    // one pair: `cur` holds its source pieces, the next pair's are requested into `nxt`
    auto one_pair = [&](auto ns_c, auto mode_c, int k, const Q16 (&cur)[VIEWS_SLOTS], Q16 (&nxt)[VIEWS_SLOTS]) {
        {
            uint4* tl4 = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(&tile4[0][0]) + buf_bytes);
            stage1(ns_c, mode_c, k, cur, tl4);
            // LDS position of rot column c0 within its row's first item: 0..3, from the yaw's shift
            uint32_t soff = buf_bytes + 4u * (((uint32_t)__builtin_amdgcn_readlane((int)X.cw0, k) >> 20) & 3u);
            asm volatile("" : "+s"(soff));  // one scalar: keeps the buffer base out of separate vector adds
#ifndef P2P_ABLATE_BARRIER
            __syncthreads();
#endif
            const unsigned char* tl = reinterpret_cast<const unsigned char*>(&tile4[0][0]);
            uint32_t ta[PXT][4];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const uint32_t* up = reinterpret_cast<const uint32_t*>(tl + (tap_up[j] + soff));
                const uint32_t* lo = reinterpret_cast<const uint32_t*>(tl + (tap_lo[j] + soff));
                ta[j][0] = up[0];
                ta[j][1] = up[1];
                ta[j][2] = lo[0];
                ta[j][3] = lo[1];
            }
            // the next pair's pieces (the last pair asks for its own again: no branch on the memory path); asking
            // for them a whole pair earlier, before stage 1, changes nothing (92.9 vs 93.0 us): not latency-bound
            const int kn = k + 1 < nplain ? k + 1 : k;
#ifndef P2P_ABLATE_LOADS
            load_pieces(ns_c, kn, nxt);
#else
            (void)kn;
#pragma unroll
            for (int sl = 0; sl < VIEWS_SLOTS; ++sl)
                nxt[sl] = cur[sl];
#endif
            uint32_t pix[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                pix[j] = blend4_packed(ta[j][0], ta[j][1], ta[j][2], ta[j][3], tw[j]);
            const int pair = X.pair0 + (int)((uint32_t)__builtin_amdgcn_readlane(X.cw3, k) >> 26);
#ifdef P2P_ABLATE_STORES
            if (pix[0] == 0x12345678u && pix[PXT - 1] == 0x9ABCDEF0u)
#endif
            {
                uint8_t* O = out + ((size_t)pair * P.n_pitch + G.pitch_i) * view_bytes;  // [pano][yaw][pitch][oh][ow][3]
                // aux 2 = nt: the views are written once and not read by this kernel, they should not displace
                // the panorama from the caches
#ifndef P2P_DPP_STORES
#pragma unroll
                for (int j = 0; j < PXT; ++j)
                    stg[j * 64 + ln] = pix[j];
                // DS operations of one wave execute in order: the read below sees the writes above
                const uint4 v = *reinterpret_cast<const uint4*>(stg + stg_rd);
                u32x3 o;
                o.x = __builtin_amdgcn_perm(v.y, v.x, 0x04020100u);  // B0 G0 R0 B1
                o.y = __builtin_amdgcn_perm(v.z, v.y, 0x05040201u);  // G1 R1 B2 G2
                o.z = __builtin_amdgcn_perm(v.w, v.z, 0x06050402u);  // R2 B3 G3 R3
#ifdef P2P_ABLATE_STORES2
                if (o.x == 0x12345678u && o.z == 0x9ABCDEF0u)
#endif
                __builtin_amdgcn_raw_buffer_store_b96(o, __builtin_amdgcn_make_buffer_rsrc(O, 0, (int)view_bytes, 0x00020000),
                                                      (int)out_off12, 0, P2P_STORE_AUX);
#else
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(O, 0, (int)view_bytes, 0x00020000);
#pragma unroll
                for (int j = 0; j < PXT; ++j) {
                    // row_shl:1 -- every lane gets its right neighbour's pixel (lane 15 of a row: unused)
                    const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)pix[j], 0x101, 0xF, 0xF, true);
                    const uint32_t dw = __builtin_amdgcn_perm(nb, pix[j], out_sel);
#ifdef P2P_ABLATE_STORES2
                    if (dw == 0x12345678u)
#endif
                    __builtin_amdgcn_raw_buffer_store_b32(dw, rsrc, (int)out_off[j], 0, P2P_STORE_AUX);
                }
#endif
            }
            buf_bytes ^= (uint32_t)sizeof(tile4[0]);
        }
    };
    // the pieces ping-pong between two register sets (pairs two at a time), so nothing is copied per pair
    auto tight = [&](auto ns_c, auto mode_c, int kbeg, int kend) {
        int k = kbeg;
        for (; k + 1 < kend; k += 2) {
            one_pair(ns_c, mode_c, k, qc, qn);
            one_pair(ns_c, mode_c, k + 1, qn, qc);
        }
        if (k < kend) {
            one_pair(ns_c, mode_c, k, qc, qn);
#pragma unroll
            for (int sl = 0; sl < VIEWS_SLOTS; ++sl)
                qc[sl] = qn[sl];
        }
    };
    auto run_ns = [&](auto ns_c) {
        load_pieces(ns_c, 0, qc);
        // Inside the loops the pieces of pair k + 1 are followed by the store of pair k, so "pieces landed" is
        // vmcnt(1).  Entering the first loop straight after the first loads the compiler would have to assume
        // vmcnt(0) for both paths.  One store that writes nothing (a buffer store through a descriptor of zero
        // records: counted like any store, dropped by the hardware) gives both paths the same shape.
        __builtin_amdgcn_sched_barrier(0);
#ifndef P2P_DPP_STORES
        __builtin_amdgcn_raw_buffer_store_b32(0u, __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000), 0, 0, 0);
#else
#pragma unroll
        for (int j = 0; j < PXT; ++j)  // as many as a pair issues
            __builtin_amdgcn_raw_buffer_store_b32(0u, __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000), 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        tight(ns_c, std::integral_constant<int, 0>{}, 0, X.n1);
        tight(ns_c, std::integral_constant<int, 1>{}, X.n1, X.n2);
        tight(ns_c, std::integral_constant<int, 2>{}, X.n2, X.n3);
    };
    static_assert(VIEWS_SLOTS == 2 || VIEWS_SLOTS == 3, "dispatch below");
    if (ns_wave == 0)
        run_ns(std::integral_constant<int, 0>{});
    else if (ns_wave == 1)
        run_ns(std::integral_constant<int, 1>{});
    else if (VIEWS_SLOTS == 2 || ns_wave == 2)
        run_ns(std::integral_constant<int, 2>{});
    else
        run_ns(std::integral_constant<int, VIEWS_SLOTS>{});
}

// ---------------------------------------------------------------------------------------------
// Rest kernel body: same arithmetic, every case distinction.
// ---------------------------------------------------------------------------------------------
struct PairCtx {      // uniform per (piece, pair)
    bool fast;        // LDS scheme applies (the yaw row is a circular shift)
    bool per_column;  // per-column weights (f4tab) instead of one f
    int joff;         // LDS position of rot column c0 within its row's first item
    uint32_t goff;    // byte offset of the first item of a footprint row within a source row
    uint32_t wrap_g;  // items with g >= wrap_g wrap to the start of the row
    uint32_t f;       // uniform weight
    int cf0;          // rot column of source column 4 * g0
    int yaw_i;
    int pano;         // panorama index
    int korig;        // pair index inside the chunk (its output slot is pair0 + korig)
};

// DIRECT = true: the body of remap_views_direct_kernel -- a piece the plan marks for direct gathers, a few pairs
// per workgroup; false: remap_views_rest_kernel -- the general loop over the LDS-scheme pieces.
template <int PXT, bool DIRECT>
__device__ __forceinline__ void draw_rest(
    const ViewsParams& P, const uint8_t* __restrict__ src, const uint32_t* __restrict__ ytab,
    const YawDesc* __restrict__ ydesc, const uint32_t* __restrict__ f4tab, uint8_t* __restrict__ out,
    const PieceHdr h, const uint32_t* __restrict__ pxw, const uint32_t* __restrict__ itw,
    uint4 (*tile4)[LDS_ITEMS_CAP])
{
    const int t = threadIdx.x;
    const PieceGeo G = piece_geo(h, t);
    const bool main_draws_plain = tight_piece<PXT>(G, P);
    const int px = G.x0 + G.col, py0 = G.y0 + G.row0;
    bool inside[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j)
        inside[j] = G.row0 + j * G.rstep < G.h && px < P.ow && py0 + j * G.rstep < P.oh;

    // output addressing: 4 horizontally adjacent pixels = 12 bytes = 3 aligned dwords
    const int lane4 = t & 3;
    const bool fast_store = (P.ow & 3) == 0;
    const size_t view_bytes = (size_t)P.oh * P.ow * 3;
    const uint32_t pix_off = (uint32_t)(((size_t)py0 * P.ow + px) * 3);  // < 3 * 32766^2 < 2^32
    const uint32_t pix_step = (uint32_t)G.rstep * (uint32_t)P.ow * 3u;
    // dword lane4 of the 12 bytes P0 P1 P2 P3: bytes of the own pixel (0-2) and of the next lane's (4-6)
    const uint32_t store_sel = lane4 == 0 ? 0x04020100u : (lane4 == 1 ? 0x05040201u : 0x06050402u);

    auto store_pixels = [&](int pair, const uint32_t (&pix)[PXT]) {
        uint8_t* O = out + ((size_t)pair * P.n_pitch + G.pitch_i) * view_bytes;  // [pano][yaw][pitch][oh][ow][3]
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const uint32_t off = pix_off + (uint32_t)j * pix_step;
            if (fast_store) {
                // lanes 4k..4k+3 hold pixels P0..P3; lanes with lane4 < 3 emit dword lane4 of the 12 bytes
                // neighbour lane's pixel: row_shl:1 DPP (lane4 groups never straddle a 16-lane row)
                uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pix[j], 0x101, 0xF, 0xF, true);
                uint32_t dw = __builtin_amdgcn_perm(nxt, pix[j], store_sel);
                uint32_t voff = off + (uint32_t)lane4;
                asm volatile("" : "+v"(voff));
                if (inside[j] && lane4 < 3)
                    __builtin_nontemporal_store(dw, reinterpret_cast<uint32_t*>(O + voff));
            } else if (inside[j]) {
                uint8_t* o = O + off;
                o[0] = (uint8_t)pix[j];
                o[1] = (uint8_t)(pix[j] >> 8);
                o[2] = (uint8_t)(pix[j] >> 16);
            }
        }
    };

    // ---- direct path: the quantised coordinates come from the plan's coordinate dump ----
    struct DirectPx {
        int ix[PXT], iy[PXT];
        uint32_t fx[PXT], fy[PXT];
        bool live[PXT];
    };
    auto load_direct = [&](DirectPx& d) {
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            int2 c = make_int2(INT32_MIN, INT32_MIN);
            if (inside[j])
                c = P.coords[((size_t)G.pitch_i * P.oh + (py0 + j * G.rstep)) * P.ow + px];
            d.ix[j] = sat_short(c.x >> 5);
            d.iy[j] = sat_short(c.y >> 5);
            d.fx[j] = (uint32_t)c.x & 31u;
            d.fy[j] = (uint32_t)c.y & 31u;
            // A pixel contributes only if its 2x2 footprint touches the panorama (BORDER_CONSTANT 0: cv::remap
            // writes borderValue when sx >= w || sx+1 < 0 || sy >= h || sy+1 < 0)
            const bool inrange = inside[j] && d.ix[j] >= -1 && d.iy[j] >= -1 && d.ix[j] < P.pw && d.iy[j] < P.ph;
            // other border modes (legacy entry point, L:179) resolve every tap to some pixel
            d.live[j] = P.border == 0 ? inrange : inside[j];
        }
    };
    auto direct_pixels = [&](const DirectPx& d, const uint8_t* __restrict__ S, int yi, uint32_t (&pix)[PXT]) {
        // same arithmetic, taps gathered from global memory through the packed yaw table
        const uint32_t* __restrict__ T = ytab + (size_t)yi * P.pw;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            pix[j] = 0;
            if (d.live[j] && P.border != 0) {
                const int xa = border_interpolate(d.ix[j], P.pw, P.border), xb = border_interpolate(d.ix[j] + 1, P.pw, P.border);
                const int ya = border_interpolate(d.iy[j], P.ph, P.border), yb = border_interpolate(d.iy[j] + 1, P.ph, P.border);
                const uint8_t* row0p = S + (size_t)ya * P.src_pitch;
                const uint8_t* row1p = S + (size_t)yb * P.src_pitch;
                const uint32_t t0 = T[xa], t1 = T[xb];
                pix[j] = blend4(rot_pixel(row0p, t0), rot_pixel(row0p, t1), rot_pixel(row1p, t0), rot_pixel(row1p, t1),
                                d.fx[j], d.fy[j]);
            } else if (d.live[j]) {
                const bool c0in = d.ix[j] >= 0, c1in = d.ix[j] + 1 < P.pw, r0in = d.iy[j] >= 0, r1in = d.iy[j] + 1 < P.ph;
                const uint8_t* row0p = S + (ptrdiff_t)d.iy[j] * P.src_pitch;
                const uint8_t* row1p = row0p + P.src_pitch;
                const uint32_t t0 = c0in ? T[d.ix[j]] : 0u, t1 = c1in ? T[d.ix[j] + 1] : 0u;
                uint32_t a = (c0in && r0in) ? rot_pixel(row0p, t0) : 0u;
                uint32_t b = (c1in && r0in) ? rot_pixel(row0p, t1) : 0u;
                uint32_t c = (c0in && r1in) ? rot_pixel(row1p, t0) : 0u;
                uint32_t dd = (c1in && r1in) ? rot_pixel(row1p, t1) : 0u;
                pix[j] = blend4(a, b, c, dd, d.fx[j], d.fy[j]);
            }
        }
    };

    if constexpr (DIRECT) {
        if (G.mode == 1)
            return;
        DirectPx d;
        load_direct(d);
        const int pair0 = blockIdx.z * P.direct_ppb;
        int pair1 = pair0 + P.direct_ppb;
        if (pair1 > P.n_panos * P.n_yaw)
            pair1 = P.n_panos * P.n_yaw;
        int pano_i = pano_of_pair(P, pair0);
        int yaw_i = pair0 - pano_i * P.n_yaw;
        for (int pair = pair0; pair < pair1; ++pair) {
            uint32_t pix[PXT];
            direct_pixels(d, src + (size_t)pano_i * P.pano_stride, yaw_i, pix);
            store_pixels(pair, pix);
            if (++yaw_i == P.n_yaw) {
                yaw_i = 0;
                ++pano_i;
            }
        }
        return;
    }
    if (G.mode != 1)
        return;  // remap_views_direct_kernel's

    // ---- LDS scheme, general loop ----
    const bool listed = P.n_rest_pairs > 0;
    const PairCtxs X = listed ? pair_contexts<true>(P, ydesc, h.c0, h.c1, t) : pair_contexts<false>(P, ydesc, h.c0, h.c1, t);
    const int kfirst = main_draws_plain ? X.n3 : 0;  // the main kernel has classes 0..3 of its pieces
    if (kfirst >= X.npairs)
        return;
    uint32_t tap_up[PXT], tap_lo[PXT];
    TapWeights tw[PXT];
    decode_px<PXT>(pxw, t, tap_up, tap_lo, tw);
    uint32_t slot_off[VIEWS_SLOTS], slot_g[VIEWS_SLOTS];
    decode_items(itw, t, G.n_items, P.src_pitch, slot_off, slot_g);
    const int wave_base = __builtin_amdgcn_readfirstlane(t & ~63);
    const int n_items = G.n_items;
    const uint32_t row_bytes = 3u * (uint32_t)P.pw;

    auto pair_ctx = [&](int k) {
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)X.cw0, k);
        const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)X.cw1, k);
        PairCtx c;
        c.goff = w0 & 0xFFFFFu;
        c.joff = (int)((w0 >> 20) & 3u);
        c.fast = (w0 >> 22) & 1u;
        c.per_column = (w0 >> 23) & 1u;
        c.f = w0 >> 24;
        c.wrap_g = w1 & 0xFFFFu;
        c.yaw_i = (int)(w1 >> 16);
        c.cf0 = __builtin_amdgcn_readlane(X.cw2, k);
        const int w3 = __builtin_amdgcn_readlane(X.cw3, k);
        c.pano = w3 & 0x3FFFFFF;
        c.korig = (int)((uint32_t)w3 >> 26);
        return c;
    };
    auto issue_loads = [&](const PairCtx& pc, const uint8_t* __restrict__ S, Q16 (&q)[VIEWS_SLOTS],
                           uint32_t (&fw)[VIEWS_SLOTS]) {
#pragma unroll
        for (int k = 0; k < VIEWS_SLOTS; ++k) {
            if (wave_base + k * VIEWS_BLOCK < n_items) {  // a wave runs slot k only if its first lane has an item there
                uint32_t off = slot_off[k] + pc.goff;
                if (slot_g[k] >= pc.wrap_g)
                    off -= row_bytes;
                q[k] = *reinterpret_cast<const Q16*>(S + off);
                if (pc.per_column) {
                    // rot column of the item's first pixel: its source column - s (mod pw)
                    int cf = 4 * (int)slot_g[k] + pc.cf0;
                    if (slot_g[k] >= pc.wrap_g)
                        cf -= P.pw;
                    if (cf < 0)
                        cf += P.pw;
                    fw[k] = f4tab[(size_t)pc.yaw_i * P.pw + cf];
                }
            }
        }
    };

    PairCtx pc = pair_ctx(kfirst);
    Q16 q[VIEWS_SLOTS];
    uint32_t fw[VIEWS_SLOTS];
    if (pc.fast)
        issue_loads(pc, src + (size_t)pc.pano * P.pano_stride, q, fw);
    int buf = 0;
    for (int ki = kfirst; ki < X.npairs; ++ki) {
        const uint8_t* __restrict__ S = src + (size_t)pc.pano * P.pano_stride;
        const int cur_yaw = pc.yaw_i;
        const int pair = listed ? pc.pano * P.n_yaw + pc.yaw_i : X.pair0 + pc.korig;
        const bool has_next = ki + 1 < X.npairs;
        uint32_t pix[PXT];

        if (pc.fast) {
            uint4* tl4 = tile4[buf];
#pragma unroll
            for (int k = 0; k < VIEWS_SLOTS; ++k) {
                if (wave_base + k * VIEWS_BLOCK < n_items) {
                    const uint32_t p0 = q[k].d[0];
                    const uint32_t p1 = __builtin_amdgcn_alignbyte(q[k].d[1], q[k].d[0], 3);
                    const uint32_t p2 = __builtin_amdgcn_alignbyte(q[k].d[2], q[k].d[1], 2);
                    const uint32_t p3 = __builtin_amdgcn_alignbyte(q[k].d[3], q[k].d[2], 1);
                    const uint32_t p4 = q[k].d[3];
                    uint4 o;
                    if (pc.per_column) {
                        const uint32_t f0 = fw[k] & 0xFFu, f1 = (fw[k] >> 8) & 0xFFu,
                                       f2 = (fw[k] >> 16) & 0xFFu, f3 = fw[k] >> 24;
                        o.x = rot_blend2(p0, p1, f0, 32u - f0);
                        o.y = rot_blend2(p1, p2, f1, 32u - f1);
                        o.z = rot_blend2(p2, p3, f2, 32u - f2);
                        o.w = rot_blend2(p3, p4, f3, 32u - f3);
                    } else if (pc.f != 0) {
                        const uint32_t f8 = 8u * pc.f, g8 = 256u - f8;
                        const uint32_t m0 = p0 & 0x00FF00FFu, n0 = p0 & 0x0000FF00u;
                        const uint32_t m1 = p1 & 0x00FF00FFu, n1 = p1 & 0x0000FF00u;
                        const uint32_t m2 = p2 & 0x00FF00FFu, n2 = p2 & 0x0000FF00u;
                        const uint32_t m3 = p3 & 0x00FF00FFu, n3 = p3 & 0x0000FF00u;
                        const uint32_t m4 = p4 & 0x00FF00FFu, n4 = p4 & 0x0000FF00u;
                        o.x = rot_blend8(m0, n0, m1, n1, f8, g8);
                        o.y = rot_blend8(m1, n1, m2, n2, f8, g8);
                        o.z = rot_blend8(m2, n2, m3, n3, f8, g8);
                        o.w = rot_blend8(m3, n3, m4, n4, f8, g8);
                    } else {
                        // whole-column yaw shift (e.g. multiples of 45 degrees on 8192 columns):
                        // stage 1 is a copy, ((32*a + 0*b + 16) >> 5) == a
                        o.x = p0 & 0x00FFFFFFu;
                        o.y = p1 & 0x00FFFFFFu;
                        o.z = p2 & 0x00FFFFFFu;
                        o.w = p3 & 0x00FFFFFFu;
                    }
                    tl4[t + k * VIEWS_BLOCK] = o;
                }
            }
            const uint32_t boff = 4u * (uint32_t)pc.joff;
            __syncthreads();
            // the 2x2 taps of this thread's pixels
            const unsigned char* tl = reinterpret_cast<const unsigned char*>(tl4);
            uint32_t ta[PXT][4];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const uint32_t* up = reinterpret_cast<const uint32_t*>(tl + (tap_up[j] + boff));
                const uint32_t* lo = reinterpret_cast<const uint32_t*>(tl + (tap_lo[j] + boff));
                ta[j][0] = up[0];
                ta[j][1] = up[1];
                ta[j][2] = lo[0];
                ta[j][3] = lo[1];
            }
            // the next pair's source loads go out now; their latency hides behind stage 2
            if (has_next) {
                pc = pair_ctx(ki + 1);
                if (pc.fast)
                    issue_loads(pc, src + (size_t)pc.pano * P.pano_stride, q, fw);
            }
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                pix[j] = blend4_packed(ta[j][0], ta[j][1], ta[j][2], ta[j][3], tw[j]);
            buf ^= 1;  // the next pair writes the other buffer; its readers are past this barrier
        } else {
            DirectPx d;
            load_direct(d);
            direct_pixels(d, S, cur_yaw, pix);
            if (has_next) {
                pc = pair_ctx(ki + 1);
                if (pc.fast)
                    issue_loads(pc, src + (size_t)pc.pano * P.pano_stride, q, fw);
            }
        }
        store_pixels(pair, pix);
    }
}

// ---------------------------------------------------------------------------------------------
// kernels: the pieces of split tiles sit in front of the tile workgroups (blockIdx.x < P.plan_gx), so that
// the few long-running ones overlap with the bulk instead of trailing it
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VIEWS_BLOCK, VIEWS_WAVES_PER_SIMD) void remap_views_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, const YawDesc* __restrict__ ydesc, uint8_t* __restrict__ out,
    const PieceHdr* __restrict__ hdr_main, const uint32_t* __restrict__ px_main, const uint32_t* __restrict__ items_main,
    const PieceHdr* __restrict__ hdr_x, const uint32_t* __restrict__ px_x, const uint32_t* __restrict__ items_x)
{
    __shared__ uint4 tile4[2][LDS_ITEMS_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t stage[(VIEWS_BLOCK / 64) * VIEWS_PXT * 64];  // a dword per pixel
    if ((int)blockIdx.x < P.plan_gx) {
        const int ei = (int)blockIdx.y * P.plan_gx + (int)blockIdx.x;
        if (ei >= P.x_n)
            return;
        const PieceHdr h = hdr_x[ei];
        draw_tight<XTRA_PXT>(P, src, ydesc, out, h, px_x + (size_t)h.px_block * (VIEWS_BLOCK * XTRA_PXT),
                             items_x + (size_t)h.item_block * LDS_ITEMS_CAP, tile4, stage);
        return;
    }
    const int bx = (int)blockIdx.x - P.plan_gx, gx = (int)gridDim.x - P.plan_gx;
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const int chunk = gx >> 3;  // gx == 8 * ceil(tiles / 8)
    const int tile_id = (bx & 7) * chunk + (bx >> 3);
    if (tile_id >= tiles)
        return;
    // heaviest views first (the host orders pitch_order by |pitch - 90| descending): a smoother tail
    const int pitch_i = P.pitch_order[blockIdx.y];
    const PieceHdr h = hdr_main[(size_t)pitch_i * tiles + tile_id];
    draw_tight<VIEWS_PXT>(P, src, ydesc, out, h, px_main + (size_t)h.px_block * (VIEWS_BLOCK * VIEWS_PXT),
                          items_main + (size_t)h.item_block * LDS_ITEMS_CAP, tile4, stage);
}

__global__ __launch_bounds__(VIEWS_BLOCK) void remap_views_rest_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, const uint32_t* __restrict__ ytab,
    const YawDesc* __restrict__ ydesc, const uint32_t* __restrict__ f4tab, uint8_t* __restrict__ out,
    const PieceHdr* __restrict__ hdr_main, const uint32_t* __restrict__ px_main, const uint32_t* __restrict__ items_main,
    const PieceHdr* __restrict__ hdr_x, const uint32_t* __restrict__ px_x, const uint32_t* __restrict__ items_x)
{
    __shared__ uint4 tile4[2][LDS_ITEMS_CAP];
    if ((int)blockIdx.x < P.plan_gx) {
        const int ei = (int)blockIdx.y * P.plan_gx + (int)blockIdx.x;
        if (ei >= P.x_n)
            return;
        const PieceHdr h = hdr_x[ei];
        draw_rest<XTRA_PXT, false>(P, src, ytab, ydesc, f4tab, out, h, px_x + (size_t)h.px_block * (VIEWS_BLOCK * XTRA_PXT),
                            items_x + (size_t)h.item_block * LDS_ITEMS_CAP, tile4);
        return;
    }
    const int bx = (int)blockIdx.x - P.plan_gx, gx = (int)gridDim.x - P.plan_gx;
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const int chunk = gx >> 3;
    const int tile_id = (bx & 7) * chunk + (bx >> 3);
    if (tile_id >= tiles)
        return;
    const int pitch_i = P.pitch_order[blockIdx.y];
    const PieceHdr h = hdr_main[(size_t)pitch_i * tiles + tile_id];
    if ((h.mode_items & 3u) == 0u)
        return;  // a split tile: drawn by its pieces
    draw_rest<VIEWS_PXT, false>(P, src, ytab, ydesc, f4tab, out, h, px_main + (size_t)h.px_block * (VIEWS_BLOCK * VIEWS_PXT),
                         items_main + (size_t)h.item_block * LDS_ITEMS_CAP, tile4);
}

// One workgroup per (direct-gather piece of the plan's list, chunk of pairs): the few pieces with a pole or the
// panorama's border inside.  Launched over the whole grid (as part of the rest kernel, round 2's first version)
// the thousands of workgroups with nothing to do cost more than the gathers.
#ifndef P2P_DIRECT_WAVES
#define P2P_DIRECT_WAVES 6  // measured 2 / 4 / 5 / 6 / 8 on the reference CLI's default view set: 108 / 104 / 102 / 101 / 110 us
#endif
__global__ __launch_bounds__(VIEWS_BLOCK, P2P_DIRECT_WAVES) void remap_views_direct_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, const uint32_t* __restrict__ ytab, uint8_t* __restrict__ out,
    const PieceHdr* __restrict__ hdr_main, const PieceHdr* __restrict__ hdr_x, const uint32_t* __restrict__ direct_list)
{
    const uint32_t id = direct_list[blockIdx.x];
    if (id & 0x80000000u)
        draw_rest<XTRA_PXT, true>(P, src, ytab, nullptr, nullptr, out, hdr_x[id & 0x7FFFFFFFu], nullptr, nullptr, nullptr);
    else
        draw_rest<VIEWS_PXT, true>(P, src, ytab, nullptr, nullptr, out, hdr_main[id], nullptr, nullptr, nullptr);
}

// which = 0: the main kernel, 1: the rest, 2: the direct-gather pieces (the three write disjoint pixels; the host
// launches the last two only when the plan or the yaw tables have something for them)
hipError_t launch_remap_views(const ViewsParams& P, int which, hipStream_t st)
{
    if (which == 2) {
        const int np = P.n_panos * P.n_yaw;
        const dim3 grid(P.n_direct, 1, (np + P.direct_ppb - 1) / P.direct_ppb);
        hipLaunchKernelGGL(remap_views_direct_kernel, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ytab, P.out, P.hdr_main,
                           P.hdr_x, P.direct_list);
        return hipGetLastError();
    }
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const int n_pairs = P.n_panos * P.n_yaw;
    int zblocks = (n_pairs + P.pairs_per_block - 1) / P.pairs_per_block;
    if (which == 1 && P.n_rest_pairs > 0)
        zblocks = (P.n_rest_pairs + P.rest_ppb - 1) / P.rest_ppb;
    // 8 XCDs, each a contiguous run of tiles; P.plan_gx is a multiple of 8 too
    const dim3 grid(P.plan_gx + 8 * ((tiles + 7) / 8), P.n_pitch, zblocks);
    if (which == 0)
        hipLaunchKernelGGL(remap_views_kernel, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ydesc, P.out, P.hdr_main,
                           P.px_main, P.items_main, P.hdr_x, P.px_x, P.items_x);
    else
        hipLaunchKernelGGL(remap_views_rest_kernel, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ytab, P.ydesc, P.f4tab,
                           P.out, P.hdr_main, P.px_main, P.items_main, P.hdr_x, P.px_x, P.items_x);
    return hipGetLastError();
}

}  // namespace p2p
