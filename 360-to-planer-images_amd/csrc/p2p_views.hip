// p2p_views.hip -- the hot kernel: both cv2.remap stages of every (panorama, yaw, pitch) view in one launch
//   cv2.remap x2       P:192-199, P:212-218 -> remap_views_kernel (both stages fused, fixed point),
//                                            remap_views_gather_kernel, remap_views_rest_kernel and remap_views_table_kernel (the odd
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
namespace P2P_SHAPE_NS {

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
// The view kernels.  One workgroup = one tile of TILE_W x TILE_H output pixels of one pitch view (VIEWS_PXT pixels
// per thread).  The plan pass (p2p_plan.hip) has already worked out everything that depends on the maps only, so a
// workgroup starts with a handful of loads.  Two schemes:
//
// LDS scheme (mode 1 tiles: the tile's footprint in the yaw-resampled panorama fits an LDS buffer).  Set-up: the
// tile header (scalar), one dword per pixel (LDS offsets of its 2x2 taps + the two 5-bit weights) and one dword per
// footprint item (rot row, 4-pixel group).  Then, per (panorama, yaw) pair of its chunk:
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
// Gather scheme (mode 2 tiles: strong minification -- the reference CLI's default 800 x 800 views of an 8K panorama
// read 2.6 source pixels per output pixel --, a pole inside the tile, taps on the panorama's border).  No LDS tile,
// no barrier: per pixel and pair two 12-byte loads fetch the three source pixels under the two rot taps of the upper
// and of the lower row, stage 1 blends just those four rot pixels in registers, stage 2 as above.  The cost per
// output pixel does not depend on the footprint.
//
// Four kernels share this arithmetic:
//   remap_views_kernel         mode 1 tiles x yaws that are plain shifts -- nothing but three branch-free loops
//                              (copy / blend / blend with the clipped column patched), no spills;
//   remap_views_gather_kernel  mode 2 tiles x yaws that are plain shifts, BORDER_CONSTANT
//                              -- three branch-free loops again (copy / blend / seam inside the tile);
//   remap_views_rest_kernel    mode 1 tiles x the job's odd pairs, the general LDS loop with its case distinctions:
//                              yaws with per-column weights, yaw rows that are not a shift (gathered per pixel);
//   remap_views_table_kernel   mode 2 tiles through the packed yaw table and cv::borderInterpolate: every pair when
//                              the gather kernel does not apply (the legacy tool's border modes), else the odd pairs.
// Device view rows are padded to whole 4-pixel groups (ViewsParams::out_row), so every lane's 12 bytes are
// dword-aligned for any view width; the host copies rows out.
// The kernels write disjoint pixels; all but the first are launched (same stream, before it) only when the plan or
// the yaw tables have something for them.
//
// Addressing: a tile's position, its pitch view and its table slots follow from its slot number (the workgroup index,
// or a work-list entry clamped to the plan's slots).  Every offset
// that comes out of a table (source offsets from items, coordinates and yaw tables; pair lists; tile lists) is either
// clamped or goes through a buffer descriptor with the exact extent, so no content of the tables can take a load or
// a store outside its buffer (p2p_audit.h; the -DP2P_AUDIT build records every such event).
// ---------------------------------------------------------------------------------------------

// the tiles the main kernel draws: LDS scheme (device view rows are whole 12-byte groups for every width: ViewsParams::out_row)
__device__ __forceinline__ bool tight_tile(const TileGeo& g, const ViewsParams& P)
{
    return g.mode == 1;
}

// ---- per-pair contexts: lane k of every wave works out pair pair0 + k once; the loops read them back with
// v_readlane, so no descriptor load sits on the per-pair critical path.  Sorted by class across the lanes:
//   0  whole-column shift (stage 1 is a copy), footprint inside one pass of the source row     } tight loops
//   1  whole-column shift, footprint across the source row's end (items wrap to its start)      }
//   2  one blend weight for the whole tile, footprint inside one pass of the row                }
//   3  one blend weight, footprint across the row's end -- where the column P:105 clips to      }
//      pw - 1 lives: that one rot pixel is a copy instead
//   4  per-column weights (a shift fraction within float noise of a rounding tie: 6 of the 360 one-degree
//      yaws on 8192 columns) or a caller row that is not a shift -> general loop of the rest kernel
//   5  the job does not draw this (yaw, pitch) view (p2p_job_set_view_mask): nobody's
struct PairCtxs {
    uint32_t cw0, cw1;
    int cw2, cw3;
    int n0, n1, n2, n3, n4;  // pairs of class 0, of classes 0..1, 0..2, 0..3, 0..4
    int npairs, pair0;
};

// the chunk of pairs a workgroup loops over: chunk * ppb ..., of all the job's pairs or of its odd-pair list.
// Tile grids run the tiles fastest and then, for ONE panorama, the chunks of a pitch view before the next pitch view
// (tile, chunk, pitch view): the view's plan tables and its band of the panorama are still in the Infinity Cache when
// the next chunk wants them (config 4, five pitch views of 113 MB of tables each: 7.5 ms instead of 8.0).  With SEVERAL
// resident panoramas the chunks run outermost (tile, pitch view, chunk): a chunk's one or two panoramas serve all
// their pitch views from the cache before the next ones are touched (config 3's share of 8: 841 vs 898 us).
// List grids (gather / table kernels) are (tile of the list, 1, chunk).
__device__ __forceinline__ int tile_grid_chunk(const ViewsParams& P) { return (int)(P.chunk_outer ? blockIdx.z : blockIdx.y); }
__device__ __forceinline__ int tile_grid_pitch_block(const ViewsParams& P) { return (int)(P.chunk_outer ? blockIdx.y : blockIdx.z); }
__device__ __forceinline__ int list_grid_chunk() { return (int)blockIdx.z; }

__device__ __forceinline__ void pair_chunk(const ViewsParams& P, bool list, int ppb, int chunk, int& first, int& count)
{
    const int n_pairs = list ? P.n_odd_pairs : P.n_panos * P.n_yaw;
    first = chunk * ppb;
    int last = first + ppb;
    if (last > n_pairs)
        last = n_pairs;
    count = last > first ? last - first : 0;
    if (count > 64)
        count = 64;  // one context per lane of a wave (the host never asks for more)
}

__device__ __forceinline__ int pair_of_lane(const ViewsParams& P, bool list, int first, int k, uint32_t site)
{
    int pair = first + k;
    if (list) {
        pair = (int)P.odd_pairs[first + k];
        P2P_AUD_LT(P.audit, site, pair, P.n_panos * P.n_yaw);
        const int last = P.n_panos * P.n_yaw - 1;
        pair = (unsigned)pair <= (unsigned)last ? pair : last;
    }
    return pair;
}

// LIST: the chunk's pairs come from P.odd_pairs (the rest kernel drawing only the yaws left to it) instead of being
// the contiguous run chunk * pairs_per_block ...
template <bool LIST = false>
__device__ __forceinline__ PairCtxs pair_contexts(const ViewsParams& P, const YawDesc* __restrict__ ydesc, int c0, int c1, int t, int chunk, int pitch_i,
                                                  int ppb = 0)
{
    PairCtxs X;
    pair_chunk(P, LIST, LIST ? P.rest_ppb : (ppb > 0 ? ppb : P.pairs_per_block), chunk, X.pair0, X.npairs);
    const int ngroups = P.pw >> 2;
    uint32_t cw0 = 0, cw1 = 0;
    int cw2 = 0, cw3 = 0, cls = 4;
    const int k = t & 63;
    if (k < X.npairs) {
        const int pair = pair_of_lane(P, LIST, X.pair0, k, AUD_REST_PAIR);
        cw3 = pano_of_pair(P, pair);
        const int yi = pair - cw3 * P.n_yaw;
        const YawDesc yd = ydesc[yi];
        int i_first = c0 + yd.s;
        if (i_first >= P.pw)
            i_first -= P.pw;
        const int g0 = i_first >> 2;
        const bool clamp_in = yd.c_clamp >= c0 && yd.c_clamp <= c1 + 1;
        // uniform weight unless this yaw flickers or the tile holds the column clipped to pw-1
        const bool per_column = yd.mode == 1 || clamp_in;
        cw0 = 12u * (uint32_t)g0 | (uint32_t)(i_first & 3) << 20 | (uint32_t)(yd.mode != 2) << 22 |
              (uint32_t)per_column << 23 | (uint32_t)yd.f << 24;
        cw1 = ((uint32_t)(ngroups - g0) & 0xFFFFu) | (uint32_t)yi << 16;  // (masked: a garbage descriptor must not reach the yaw field)
        cw2 = 4 * g0 - yd.s;
        cw3 |= k << 26;  // n_panos < 2^26 (host check): the chunk-local pair index rides along
        // Items reach at most group ((c1 + 2 - c0) + 7) >> 2 past the first (the taps' columns c0 .. c1 + 1, the yaw's
        // alignment 0..3, whole groups); beyond the row's last group, ngroups - g0 groups on, they wrap to its start.
        // A pair none of whose items can wrap loads its pieces without the wrap test (3 instructions per item less):
        // classes 0 and 2.  The pieces are requested one pair ahead, so the loops below run a class's pairs but the last
        // with the test-free loads (the next pair is of the same class) and the last one with the general ones.
#ifdef P2P_NO_WRAP_CLASSES
        const bool wraps = clamp_in;
#else
        const bool wraps = clamp_in || (ngroups - g0) <= (((c1 + 2 - c0) + 7) >> 2);
#endif
        cls = yd.mode != 0 ? 4 : (yd.f == 0 ? (wraps ? 1 : 0) : (wraps ? 3 : 2));
        if (!view_wanted(P, pitch_i, yi))
            cls = 5;
    }
    int cum[6];
    const int r = sort_lanes_by_class<6>(k, k < X.npairs, cls, cum);
    X.n0 = cum[0];
    X.n1 = cum[1];
    X.n2 = cum[2];
    X.n3 = cum[3];
    X.n4 = cum[4];
    X.cw0 = (uint32_t)__builtin_amdgcn_ds_permute(4 * r, (int)cw0);
    X.cw1 = (uint32_t)__builtin_amdgcn_ds_permute(4 * r, (int)cw1);
    X.cw2 = __builtin_amdgcn_ds_permute(4 * r, cw2);
    X.cw3 = __builtin_amdgcn_ds_permute(4 * r, cw3);
    return X;
}

// The same contexts from the job's table (ViewsParams::pair_ctx, written by pair_ctx_kernel with the function above): one
// 16-byte load per lane instead of 83 instructions per wave -- a tile's set-up is worth 2.4 pairs of its 12 on config 2,
// a whole pair of the 4 a band tile of the reference CLI's default set draws.  Record of lane k: cw0, cw1, cw3 of the
// k-th pair in class order; the fourth word of lane 0 holds n0 | n1 << 8 | n2 << 16 | n3 << 24, of lane 1 n4 | npairs << 8.
__device__ __forceinline__ PairCtxs pair_contexts_from_table(const ViewsParams& P, uint32_t slot, int chunk, int t, int ppb_eff)
{
    PairCtxs X;
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (chunk < P.pair_ctx_chunks)  // (a workgroup that loops over chunks may look one past the last: no pairs there)
        r = P.pair_ctx[((size_t)slot * (size_t)P.pair_ctx_chunks + (size_t)chunk) * 64u + (size_t)(t & 63)];
    X.cw0 = r.x;
    X.cw1 = r.y;
    X.cw2 = 0;  // (the rest kernel's: it works its contexts out itself)
    X.cw3 = (int)r.z;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)r.w, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)r.w, 1);
    // (whatever the table holds, the counts stay an ascending sequence within a wave's 64 lanes)
    X.n0 = (int)min(a & 0xFFu, 64u);
    X.n1 = max(X.n0, (int)min((a >> 8) & 0xFFu, 64u));
    X.n2 = max(X.n1, (int)min((a >> 16) & 0xFFu, 64u));
    X.n3 = max(X.n2, (int)min(a >> 24, 64u));
    X.n4 = max(X.n3, (int)min(b & 0xFFu, 64u));
    X.npairs = (int)min((b >> 8) & 0xFFu, 64u);
    X.pair0 = chunk * ppb_eff;
    return X;
}

// per-pixel words of the plan -> tap offsets (bytes inside one LDS buffer) and packed weights
// band_row_bytes > 0: a band tile's words -- the distance field is a live flag, the lower tap sits one LDS row further
template <int PXT>
__device__ __forceinline__ void decode_px(const uint32_t* __restrict__ pxw, int t, uint32_t (&tap_up)[PXT],
                                          uint32_t (&tap_lo)[PXT], TapWeights (&tw)[PXT], uint32_t band_row_bytes = 0u)
{
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
#ifdef P2P_ABLATE_HALF_PX  // timing experiment (wrong pixels): half the lines of the per-pixel words -- what 2-byte words would read
        uint32_t wd = pxw[(j >> 1) * VIEWS_BLOCK + t];
        asm volatile("" : "+v"(wd));  // (two pixels with one word: without this the compiler draws them once -- "8 % faster")
#else
        const uint32_t wd = pxw[j * VIEWS_BLOCK + t];
#endif
        const uint32_t dl = (wd >> PXW_UP_BITS) & ((1u << PXW_DL_BITS) - 1u);
        tap_up[j] = (wd & ((1u << PXW_UP_BITS) - 1u)) << 2;
        tap_lo[j] = tap_up[j] + (band_row_bytes ? band_row_bytes : dl << 2);
        // a pixel with no footprint in the panorama (NaN coordinate) gets weight 0 everywhere:
        // (0 + 512) >> 10 == 0, the BORDER_CONSTANT value
        const uint32_t fx = (wd >> 22) & 31u, fy = wd >> 27;
        tw[j] = tap_weights(fx, fy, dl != 0);
    }
}

// The way out of the tight kernels: a wave's pixels -> LDS (a dword per pixel) -> 4 adjacent pixels of one row per
// lane -> 12 bytes, written with a buffer store whose descriptor covers exactly this view: lanes with nothing to
// store (rows past the view, four-pixel groups past its right edge) get an offset beyond it and the hardware
// drops them.  No lane is masked off: a masked store brings a branch, and with it vmcnt(0).
struct StoreCtx {
    uint32_t* stg;       // this wave's staging dwords
    uint32_t stg_rd;     // where this lane reads its four pixels back
    uint32_t out_off12;  // byte offset of those 12 bytes inside a view (0xFFFFFFFF: nothing to store)
    int ln;
};

__device__ __forceinline__ StoreCtx store_ctx(const ViewsParams& P, const TileGeo& G, uint32_t* stage, int t)
{
    StoreCtx s;
    const int wv = t >> 6;
    s.ln = t & 63;
    s.stg = stage + wv * (VIEWS_PXT * 64);
    const int x4 = 4 * (s.ln & 15), sj = s.ln >> 4;  // the group's first pixel as a lane of this wave; which of the thread's pixels
    const int srow = ((wv * 64 + x4) >> TILE_LW) + sj * TILE_ROWSTEP, scol = (wv * 64 + x4) & (TILE_W - 1);
    const bool s_ok = sj < VIEWS_PXT && srow < TILE_H && G.y0 + srow < P.oh && G.x0 + scol < P.ow;
    s.out_off12 = s_ok ? (uint32_t)(G.y0 + srow) * (uint32_t)P.out_row + 3u * (uint32_t)(G.x0 + scol) : 0xFFFFFFFFu;
    s.stg_rd = (uint32_t)(sj * 64 + x4);
    return s;
}

// first half: the wave's pixels into its staging dwords (stg: this wave's 64 x VIEWS_PXT dwords)
__device__ __forceinline__ void stage_wave_pixels(const StoreCtx& s, uint32_t* stg, const uint32_t (&pix)[VIEWS_PXT])
{
#pragma unroll
    for (int j = 0; j < VIEWS_PXT; ++j)
        stg[j * 64 + s.ln] = pix[j];
}

// second half: four adjacent pixels of one row back out of the staging dwords (DS operations of one wave execute in
// order: the read sees the wave's own writes, also those of the pair before), 12 bytes, one store.  `records` = bytes
// of the view, or 0: a descriptor of no records, every lane dropped by the hardware
__device__ __forceinline__ uint4 read_staged_pixels(const StoreCtx& s, const uint32_t* stg)
{
    return *reinterpret_cast<const uint4*>(stg + s.stg_rd);
}

__device__ __forceinline__ u32x3 pack_staged_pixels(const uint4& v)
{
    u32x3 o;
    o.x = __builtin_amdgcn_perm(v.y, v.x, 0x04020100u);  // B0 G0 R0 B1
    o.y = __builtin_amdgcn_perm(v.z, v.y, 0x05040201u);  // G1 R1 B2 G2
    o.z = __builtin_amdgcn_perm(v.w, v.z, 0x06050402u);  // R2 B3 G3 R3
    return o;
}

__device__ __forceinline__ void store_packed_pixels(const StoreCtx& s, const u32x3& o, uint8_t* O, uint32_t records)
{
    __builtin_amdgcn_raw_buffer_store_b96(o, __builtin_amdgcn_make_buffer_rsrc(O, 0, (int)records, 0x00020000),
                                          (int)s.out_off12, 0, TILE_W == 128 ? P2P_MAIN_STORE_AUX_W128 : P2P_MAIN_STORE_AUX_W64);
}

__device__ __forceinline__ void store_staged_pixels(const StoreCtx& s, const uint4& v, uint8_t* O, uint32_t records)
{
    store_packed_pixels(s, pack_staged_pixels(v), O, records);
}

__device__ __forceinline__ void store_wave_pixels(const StoreCtx& s, const uint32_t (&pix)[VIEWS_PXT], uint8_t* O, size_t view_bytes)
{
#pragma unroll
    for (int j = 0; j < VIEWS_PXT; ++j)
        s.stg[j * 64 + s.ln] = pix[j];
    // DS operations of one wave execute in order: the read below sees the writes above
    const uint4 v = *reinterpret_cast<const uint4*>(s.stg + s.stg_rd);
    u32x3 o;
    o.x = __builtin_amdgcn_perm(v.y, v.x, 0x04020100u);  // B0 G0 R0 B1
    o.y = __builtin_amdgcn_perm(v.z, v.y, 0x05040201u);  // G1 R1 B2 G2
    o.z = __builtin_amdgcn_perm(v.w, v.z, 0x06050402u);  // R2 B3 G3 R3
#ifdef P2P_ABLATE_STORES2
    if (o.x == 0x12345678u && o.z == 0x9ABCDEF0u)
#endif
    // aux 2 = nt: the views are written once and not read by this kernel, they should not displace the panorama
    __builtin_amdgcn_raw_buffer_store_b96(o, __builtin_amdgcn_make_buffer_rsrc(O, 0, (int)view_bytes, 0x00020000),
                                          (int)s.out_off12, 0, P2P_STORE_AUX);
}

// ---------------------------------------------------------------------------------------------
// Main kernel body: the three tight loops.  Free of branches on the vector-memory path, so that the compiler's
// s_waitcnt vmcnt stay counted (with a conditional load or store in the loop it falls back to vmcnt(0), and every
// pair then waits for the previous pair's stores to be acknowledged: loads and stores retire in issue order).
// ---------------------------------------------------------------------------------------------
// BAND: a source-band tile (p2p_device.h).  Its footprint is a rectangle (G.band_r0, G.band_row_items: no item list,
// itw unused), lane t draws the 4 adjacent pixels of group t and stores them itself (no staging): grpw[t] is the byte
// offset of its 12 bytes inside the n_pitch views of a (panorama, yaw) pair.
template <bool BAND = false, bool MASKED = false>
__device__ __forceinline__ void draw_tight(
    const ViewsParams& P, const uint8_t* __restrict__ src, const YawDesc* __restrict__ ydesc,
    uint8_t* __restrict__ out, const TileGeo& G, const uint32_t* __restrict__ pxw, const uint32_t* __restrict__ itw,
    uint4 (*tile4)[LDS_ITEMS_CAP], uint32_t* stage, int chunk, int ppb = 0, const uint32_t* __restrict__ grpw = nullptr)
{
    constexpr int PXT = VIEWS_PXT;
    const int t = threadIdx.x;
    if (!BAND && !tight_tile(G, P))
        return;  // the other kernels'
    P2P_AUD_LT(P.audit, AUD_MAIN_HDR, G.n_items, LDS_ITEMS_CAP + 1);
    // A workgroup draws P.main_span consecutive chunks of pairs (1 unless the plan tables are too big to stay cached:
    // the tile's words and items are then read and decoded once for all of them instead of once per chunk).
    // Only the 128-wide shape has the loop: the 64-wide kernel sits at its 72 registers (seven workgroups per CU), and
    // the loop's live values cost it 5 spilled ones; the host asks for spans with 128-wide tiles only.
    constexpr bool SPAN_LOOP = TILE_W == 128;
    const int chunk_end = chunk + (SPAN_LOOP && P.main_span > 1 ? P.main_span : 1);
    // (band tiles: lanes of several pitch views -- no pair is skipped as a whole; unwanted views are dropped per lane below)
    // (from the job's table unless this workgroup draws a PART of the pairs -- the split tail of a list, ppb > 0)
    const bool ctx_table = P.pair_ctx != nullptr && ppb == 0;
    PairCtxs X = ctx_table ? pair_contexts_from_table(P, G.slot, chunk, t, P.pairs_per_block)
                           : pair_contexts(P, ydesc, G.c0, G.c1, t, chunk, BAND ? -1 : G.pitch_i, ppb);
    int nplain = X.n3;
    if (nplain == 0 && chunk + 1 == chunk_end)
        return;
    uint32_t tap_up[PXT], tap_lo[PXT];
    TapWeights tw[PXT];
    const uint32_t band_row_bytes = BAND ? 16u * (uint32_t)G.band_row_items : 0u;
    // band tiles: where this lane's 12 bytes go inside a pair's n_pitch views (~0: no group, the hardware drops the
    // store).  Does this wave hold any group?  A footprint-limited tile -- a minifying view set -- fills only its first
    // waves; the others produce their items, skip stage 2 and need no per-pixel words.
    uint32_t grp_off = 0xFFFFFFFFu;
    if (BAND) {
        grp_off = grpw[t];
        P2P_AUD_LT(P.audit, AUD_BAND_GRP, grp_off == 0xFFFFFFFFu ? 0u : grp_off + 11u, (uint32_t)P.n_pitch * (uint32_t)P.view_bytes);
    }
    const bool wave_draws = !BAND || __ballot(grp_off != 0xFFFFFFFFu) != 0ull;
    if (wave_draws) {
        decode_px<PXT>(pxw, t, tap_up, tap_lo, tw, band_row_bytes);
    } else {
#pragma unroll
        for (int jj = 0; jj < PXT; ++jj) {
            tap_up[jj] = tap_lo[jj] = 0u;
            tw[jj].w_up = tw[jj].w_lo = 0u;
        }
    }
#ifdef P2P_AUDIT
#pragma unroll
    for (int j = 0; j < PXT; ++j)
        P2P_AUD_RANGE(P.audit, AUD_MAIN_TAP, tap_lo[j] + 12u, 8u, sizeof(tile4[0]));
#endif
#ifdef P2P_ABLATE_CONFLICTS
    // timing experiment only (wrong pixels): every lane reads its own two dwords -- tap reads without bank conflicts
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        tap_up[j] = (uint32_t)(t & 63) * 8u + (uint32_t)j * 512u;
        tap_lo[j] = tap_up[j] + 2048u;
    }
#endif
    uint32_t slot_off[VIEWS_SLOTS], slot_g[VIEWS_SLOTS];
    if (BAND) {
        // the footprint is a rectangle: item i = (row i / row_items, group i % row_items); surplus lanes redo item 0
        const uint32_t ri = (uint32_t)G.band_row_items;
        const float ri_inv = 1.0f / (float)ri;
#pragma unroll
        for (int k = 0; k < VIEWS_SLOTS; ++k) {
            uint32_t item = (uint32_t)(t + k * VIEWS_BLOCK);
            item = item < (uint32_t)G.n_items && item < (uint32_t)LDS_ITEMS_CAP ? item : 0u;
            uint32_t row = (uint32_t)((float)item * ri_inv);  // item < 2^11: the quotient is off by one at most
            row -= row * ri > item ? 1u : 0u;
            row += (row + 1u) * ri <= item ? 1u : 0u;
            slot_g[k] = item - row * ri;
            slot_off[k] = ((uint32_t)G.band_r0 + row) * (uint32_t)P.src_pitch + 12u * slot_g[k];
        }
    } else {
        decode_items(itw, t, G.n_items, P.src_pitch, slot_off, slot_g);
    }
    const uint32_t row_bytes = 3u * (uint32_t)P.pw;
    const size_t view_bytes = P.view_bytes;
    StoreCtx SC{};
    if (!BAND)
        SC = store_ctx(P, G, stage, t);
    // band tiles, sparse view sets only: which of the chunk's pairs this lane's pitch view wants
    unsigned long long lane_want = ~0ull;
    int lane_pitch = 0;
    auto lane_wants = [&]() {  // (per chunk of pairs)
        lane_want = 0ull;
        for (int k = 0; k < X.npairs; ++k) {
            const int pair = X.pair0 + k;
            const int yi = pair - pano_of_pair(P, pair) * P.n_yaw;
            lane_want |= (unsigned long long)view_wanted(P, lane_pitch, yi) << k;
        }
    };
    if (BAND) {
        if (MASKED) {
            // (one integer division per workgroup, sparse view sets only)
            lane_pitch = grp_off == 0xFFFFFFFFu ? 0 : (int)min((uint32_t)(P.n_pitch - 1), grp_off / (uint32_t)view_bytes);
            lane_wants();
        }
    }

    const int wave_base = __builtin_amdgcn_readfirstlane(t & ~63);
    int ns_wave = 0;  // items this wave produces per pair (wave-uniform)
#pragma unroll
    for (int k = 0; k < VIEWS_SLOTS; ++k)
        ns_wave += G.n_items > wave_base + k * VIEWS_BLOCK;

    uint32_t bias_br = 0x00800080u;  // rounding of both 16-bit fields, kept in a VGPR (see vmad24)
    asm volatile("" : "+v"(bias_br));

    uint32_t buf_bytes = 0u;
    Q16 qc[VIEWS_SLOTS], qn[VIEWS_SLOTS];
    // the context words of the pair being drawn, as scalars: read from their lane once, when the pair's pieces are
    // requested (one pair ahead), and handed on -- three v_readlane_b32 per pair instead of six
    struct PairWords { uint32_t w0, w1; int w3; };
    auto pair_words = [&](int k) {
        PairWords w;
        w.w0 = (uint32_t)__builtin_amdgcn_readlane((int)X.cw0, k);
        w.w1 = (uint32_t)__builtin_amdgcn_readlane((int)X.cw1, k);
        w.w3 = __builtin_amdgcn_readlane(X.cw3, k);
        return w;
    };
    PairWords pwc = pair_words(0);
    // NW (std::true_type): the pair is of class 0 or 2, none of its items wraps
    auto load_pieces = [&](auto ns_c, auto nw_c, const PairWords& W, Q16 (&qq)[VIEWS_SLOTS]) {
        constexpr int NS = decltype(ns_c)::value;
        constexpr bool NW = decltype(nw_c)::value;
        const uint32_t w0 = W.w0;
        const uint32_t wrap_g = W.w1 & 0xFFFFu;
        // one descriptor per panorama: an item word that points outside it loads zeros instead of faulting
#ifdef P2P_ABLATE_ONE_PANO  // timing experiment (wrong pixels): every resident panorama is the first one -- what sources that never come cold from HBM would give
        const auto S = make_buf(src, (uint32_t)P.pano_stride);
#else
        const auto S = make_buf(src + (size_t)(W.w3 & 0x3FFFFFF) * P.pano_stride, (uint32_t)P.pano_stride);
#endif
        const uint32_t goff = w0 & 0xFFFFFu;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            uint32_t off = slot_off[sl] + goff;
#ifndef P2P_ABLATE_WRAP  // (timing experiment: what rows padded with a copy of their first columns would save)
            if (!NW)
                off = slot_g[sl] >= wrap_g ? off - row_bytes : off;  // items past the end of the row continue at its start
#endif
#ifdef P2P_ABLATE_LOADS2
#ifndef P2P_ABLATE_LOADS2_MASK
#define P2P_ABLATE_LOADS2_MASK 0x3FFFu
#endif
            off &= P2P_ABLATE_LOADS2_MASK;  // timing experiment (wrong pixels): every load issued, all of them hits in 16 KB (or the window asked for)
#endif
            P2P_AUD_RANGE(P.audit, AUD_MAIN_SRC, off, 16u, P.pano_stride);
            const bu32x4 q = __builtin_amdgcn_raw_buffer_load_b128(S, (int)off, 0, P2P_SRC_LOAD_AUX);
            qq[sl].d[0] = q.x; qq[sl].d[1] = q.y; qq[sl].d[2] = q.z; qq[sl].d[3] = q.w;
        }
    };
    // MODE 0: copy, 1: blend, 2: blend, and the rot pixel whose source column is pw - 1 (P:105's clip) is a copy
    auto stage1 = [&](auto ns_c, auto mode_c, const PairWords& W, const Q16 (&qq)[VIEWS_SLOTS], uint4* tl4) {
        constexpr int NS = decltype(ns_c)::value;
        constexpr int MODE = decltype(mode_c)::value;
        const uint32_t f = W.w0 >> 24;
        const uint32_t last_g = (W.w1 & 0xFFFFu) - 1u;
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
    // The way out runs ONE PAIR BEHIND (unless P2P_STORE_INLINE): a pair's pixels go into the wave's staging dwords at the
    // end of its stage 2 and stay there; they are read back, packed and stored by the NEXT pair, between its stage 1 and
    // its barrier -- where the wave waits for its LDS writes anyway, so the staging round trip (write, read, lgkmcnt(0))
    // no longer stands between stage 2 and the next pair.  The first pair "stores" through a descriptor of no records.
    // The staging dwords are not a buffer of their own (unless P2P_STAGE_OWN_LDS): a wave stages into the tile buffer that
    // is NOT being read -- it holds the pair before, which every wave has finished with at the last barrier -- and
    // there into the 64 items (1 KB) that only this wave itself writes in stage 1 (items t + sl * VIEWS_BLOCK: slot 0 of
    // its own lanes).  The pixels are read back at the top of the next pair, before the wave's stage-1 writes to the
    // same addresses (DS operations of one wave execute in order).  4 KB of LDS less per workgroup: seven per CU.
    uint8_t* pend_O = out;
    uint32_t pend_records = 0u;
    auto staging = [&](uint32_t buffer_bytes) {
#ifdef P2P_STAGE_OWN_LDS
        (void)buffer_bytes;
        return SC.stg;
#else
        return reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(&tile4[0][0]) + buffer_bytes) + __builtin_amdgcn_readfirstlane(t >> 6) * (64 * 4);
#endif
    };
    static_assert(VIEWS_PXT == 4, "a wave's staging dwords are the 64 items of its lanes' first slot");
#ifdef P2P_ABLATE_HALF_TAPS
    uint32_t ta_keep[PXT][4] = {};  // (timing experiment, wrong pixels: every second pair re-uses the taps of the pair before)
#endif
    auto one_pair = [&](auto ns_c, auto mode_c, auto nw_c, auto draws_c, int k, const Q16 (&cur)[VIEWS_SLOTS], Q16 (&nxt)[VIEWS_SLOTS]) {
        if constexpr (BAND) {
            // stage 1 -> barrier -> taps -> the next pair's pieces -> stage 2 -> 12 bytes per lane, stored from registers
            constexpr bool DRAWS = decltype(draws_c)::value;
            uint4* tl4 = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(&tile4[0][0]) + buf_bytes);
            stage1(ns_c, mode_c, pwc, cur, tl4);
            // (the lower taps: one LDS row further for every pixel of a band tile -- a second scalar, no second offset per pixel)
            uint32_t soff = buf_bytes + 4u * ((pwc.w0 >> 20) & 3u), soff_lo = soff + band_row_bytes;
            asm volatile("" : "+s"(soff), "+s"(soff_lo));
            {
                uint32_t soff_v, soff_lo_v;
                asm volatile("v_mov_b32 %0, %1" : "=v"(soff_v) : "s"(soff));
                asm volatile("v_mov_b32 %0, %1" : "=v"(soff_lo_v) : "s"(soff_lo));
                soff = soff_v;
                soff_lo = soff_lo_v;
            }
            __syncthreads();
            const unsigned char* tl = reinterpret_cast<const unsigned char*>(&tile4[0][0]);
            uint32_t ta[PXT][4];
            if (DRAWS) {
#pragma unroll
                for (int j = 0; j < PXT; ++j) {
                    const uint32_t* up = reinterpret_cast<const uint32_t*>(tl + (tap_up[j] + soff));
                    const uint32_t* lo = reinterpret_cast<const uint32_t*>(tl + (tap_up[j] + soff_lo));
                    ta[j][0] = up[0];
                    ta[j][1] = up[1];
                    ta[j][2] = lo[0];
                    ta[j][3] = lo[1];
                }
            }
            const PairWords pwn = pair_words(k + 1 < nplain ? k + 1 : k);
            load_pieces(ns_c, nw_c, pwn, nxt);
            if (DRAWS) {
                uint4 v;
                v.x = blend4_packed(ta[0][0], ta[0][1], ta[0][2], ta[0][3], tw[0]);
                v.y = blend4_packed(ta[1][0], ta[1][1], ta[1][2], ta[1][3], tw[1]);
                v.z = blend4_packed(ta[2][0], ta[2][1], ta[2][2], ta[2][3], tw[2]);
                v.w = blend4_packed(ta[3][0], ta[3][1], ta[3][2], ta[3][3], tw[3]);
                const uint32_t korig = (uint32_t)pwc.w3 >> 26;
                const int pair = X.pair0 + (int)korig;
                uint32_t off = grp_off;
                if (MASKED)
                    off = (lane_want >> korig) & 1ull ? off : 0xFFFFFFFFu;
                const uint32_t pair_bytes = (uint32_t)P.n_pitch * (uint32_t)view_bytes;  // < 2^32 (host check)
                const u32x3 o3 = pack_staged_pixels(v);
#ifdef P2P_ABLATE_STORES2  // (timing experiment: the store instruction alone skipped)
                if ((o3.x ^ o3.y ^ o3.z) == 0x12345678u)
#endif
                __builtin_amdgcn_raw_buffer_store_b96(o3,
                                                      __builtin_amdgcn_make_buffer_rsrc(out + (size_t)pair * pair_bytes, 0, (int)pair_bytes, 0x00020000),
                                                      (int)off, 0, P2P_BAND_STORE_AUX);
            }
            buf_bytes ^= (uint32_t)sizeof(tile4[0]);
            pwc = pwn;
        } else {
            uint4* tl4 = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(&tile4[0][0]) + buf_bytes);
#ifndef P2P_STORE_INLINE
            // the read-back goes out first and is consumed after stage 1 (the scheduler would otherwise pull the whole
            // chain up behind the previous pair's staging writes: the round trip this order exists to avoid)
            __builtin_amdgcn_sched_barrier(0);
            const uint4 staged = read_staged_pixels(SC, staging(buf_bytes));
            __builtin_amdgcn_sched_barrier(0);
#endif
            stage1(ns_c, mode_c, pwc, cur, tl4);
#ifndef P2P_STORE_INLINE
            __builtin_amdgcn_sched_barrier(0);
#ifdef P2P_STORE_AFTER_LOADS
            const u32x3 packed = pack_staged_pixels(staged);  // (stored behind the next pair's loads, below)
#else
            store_staged_pixels(SC, staged, pend_O, pend_records);
#endif
#endif
            // LDS position of rot column c0 within its row's first item: 0..3, from the yaw's shift
            uint32_t soff = buf_bytes + 4u * ((pwc.w0 >> 20) & 3u);
#ifdef P2P_TAP_ADD_SGPR
            asm volatile("" : "+s"(soff));  // one scalar: keeps the buffer base out of separate vector adds
#else
            // ... and that scalar in a VGPR: v_add_u32 with two vector operands issues in 2 cycles, with a scalar
            // operand in 4 (tools/ubench/valu_rates.hip) -- one v_mov and eight fast adds per pair instead of eight slow ones
            asm volatile("" : "+s"(soff));
            {
                uint32_t soff_v;
                asm volatile("v_mov_b32 %0, %1" : "=v"(soff_v) : "s"(soff));
                soff = soff_v;
            }
#endif
#ifndef P2P_ABLATE_BARRIER
            __syncthreads();
#endif
            const unsigned char* tl = reinterpret_cast<const unsigned char*>(&tile4[0][0]);
            uint32_t ta[PXT][4];
#ifdef P2P_ABLATE_HALF_TAPS
            if (&cur[0] == &qc[0]) {
#endif
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const uint32_t* up = reinterpret_cast<const uint32_t*>(tl + (tap_up[j] + soff));
                const uint32_t* lo = reinterpret_cast<const uint32_t*>(tl + (tap_lo[j] + soff));
                ta[j][0] = up[0];
                ta[j][1] = up[1];
                ta[j][2] = lo[0];
                ta[j][3] = lo[1];
            }
#ifdef P2P_ABLATE_HALF_TAPS
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                for (int q4 = 0; q4 < 4; ++q4)
                    ta_keep[j][q4] = ta[j][q4];
            } else {
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                for (int q4 = 0; q4 < 4; ++q4)
                    ta[j][q4] = ta_keep[j][q4] + (uint32_t)k;
            }
#endif
            // the next pair's pieces (the last pair asks for its own again: no branch on the memory path); asking
            // for them a whole pair earlier, before stage 1, changes nothing (92.9 vs 93.0 us): not latency-bound
            const PairWords pwn = pair_words(k + 1 < nplain ? k + 1 : k);
#ifndef P2P_ABLATE_LOADS
            load_pieces(ns_c, nw_c, pwn, nxt);
#else
#pragma unroll
            for (int sl = 0; sl < VIEWS_SLOTS; ++sl)
                nxt[sl] = cur[sl];
#endif
#if defined(P2P_STORE_AFTER_LOADS) && !defined(P2P_STORE_INLINE)
            // the store BEHIND the loads: the wait for those pieces (vmcnt counts in issue order) then does not include
            // this store's acknowledgement, only the one of the pair before
            __builtin_amdgcn_sched_barrier(0);
            store_packed_pixels(SC, packed, pend_O, pend_records);
            __builtin_amdgcn_sched_barrier(0);
#endif
            uint32_t pix[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                pix[j] = blend4_packed(ta[j][0], ta[j][1], ta[j][2], ta[j][3], tw[j]);
            const int pair = X.pair0 + (int)((uint32_t)pwc.w3 >> 26);
#ifdef P2P_STORE_INLINE
            store_wave_pixels(SC, pix, out + ((size_t)pair * P.n_pitch + G.pitch_i) * view_bytes, view_bytes);  // [pano][yaw][pitch][oh][ow][3]
#else
            stage_wave_pixels(SC, staging(buf_bytes ^ (uint32_t)sizeof(tile4[0])), pix);
            pend_O = out + ((size_t)pair * P.n_pitch + G.pitch_i) * view_bytes;  // [pano][yaw][pitch][oh][ow][3]
            pend_records = (uint32_t)view_bytes;
#endif
            buf_bytes ^= (uint32_t)sizeof(tile4[0]);
            pwc = pwn;
        }
    };
    // the pieces ping-pong between two register sets (pairs two at a time), so nothing is copied per pair
    // nw_c: the pieces requested inside the loop (those of pairs kbeg + 1 .. kend) belong to pairs without wrapping items
    auto tight = [&](auto ns_c, auto mode_c, auto nw_c, auto draws_c, int kbeg, int kend) {
        int k = kbeg;
        for (; k + 1 < kend; k += 2) {
            one_pair(ns_c, mode_c, nw_c, draws_c, k, qc, qn);
            one_pair(ns_c, mode_c, nw_c, draws_c, k + 1, qn, qc);
        }
        if (k < kend) {
            one_pair(ns_c, mode_c, nw_c, draws_c, k, qc, qn);
#pragma unroll
            for (int sl = 0; sl < VIEWS_SLOTS; ++sl)
                qc[sl] = qn[sl];
        }
    };
    auto run_ns_d = [&](auto ns_c, auto draws_c) {
        load_pieces(ns_c, std::false_type{}, pwc, qc);
        // Inside the loops the pieces of pair k + 1 are followed by the store of pair k, so "pieces landed" is
        // vmcnt(1).  Entering the first loop straight after the first loads the compiler would have to assume
        // vmcnt(0) for both paths.  One store that writes nothing (a buffer store through a descriptor of zero
        // records: counted like any store, dropped by the hardware) gives both paths the same shape.
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_buffer_store_b32(0u, __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000), 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef P2P_NO_WRAP_CLASSES
        tight(ns_c, std::integral_constant<int, 0>{}, std::false_type{}, draws_c, 0, X.n1);
        tight(ns_c, std::integral_constant<int, 1>{}, std::false_type{}, draws_c, X.n1, X.n2);
        tight(ns_c, std::integral_constant<int, 2>{}, std::false_type{}, draws_c, X.n2, X.n3);
#else
        // classes 0 | 1 copy, 2 blend, 3 blend with the clipped column; 0 and 2: no item wraps
        const int a = X.n0 > 0 ? X.n0 - 1 : 0, b = X.n2 - 1 > X.n1 ? X.n2 - 1 : X.n1;
        tight(ns_c, std::integral_constant<int, 0>{}, std::true_type{}, draws_c, 0, a);
        tight(ns_c, std::integral_constant<int, 0>{}, std::false_type{}, draws_c, a, X.n1);
        tight(ns_c, std::integral_constant<int, 1>{}, std::true_type{}, draws_c, X.n1, b);
        tight(ns_c, std::integral_constant<int, 1>{}, std::false_type{}, draws_c, b, X.n2);
        tight(ns_c, std::integral_constant<int, 2>{}, std::false_type{}, draws_c, X.n2, X.n3);
#endif
#ifndef P2P_STORE_INLINE
        if (!BAND)
            store_staged_pixels(SC, read_staged_pixels(SC, staging(buf_bytes)), pend_O, pend_records);  // the last pair's pixels
#endif
    };
    auto run_ns = [&](auto ns_c) {
        if (!BAND || wave_draws)
            run_ns_d(ns_c, std::true_type{});
        else if constexpr (BAND)
            run_ns_d(ns_c, std::false_type{});
    };
    static_assert(VIEWS_SLOTS >= 2 && VIEWS_SLOTS <= 4, "dispatch below");
    for (int ch = chunk;;) {
        if (nplain > 0) {  // (the same for every wave of the workgroup: the loops' barriers stay uniform)
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
        if (!SPAN_LOOP || ++ch >= chunk_end)
            break;
        // the next chunk of this tile: new pair contexts, everything else stands.  The tile buffers keep alternating
        // (the last pair's taps are still being read by slower waves from the buffer this wave does NOT write next).
        X = ctx_table ? pair_contexts_from_table(P, G.slot, ch, t, P.pairs_per_block)
                      : pair_contexts(P, ydesc, G.c0, G.c1, t, ch, BAND ? -1 : G.pitch_i, ppb);
        nplain = X.n3;
        if (X.npairs == 0)
            break;  // past the job's last chunk
        if (BAND && MASKED)
            lane_wants();
        pwc = pair_words(0);
        pend_records = 0u;  // (the chunk's last pair has been flushed)
    }
}

// ---------------------------------------------------------------------------------------------
// Gather kernel body (mode 2 tiles x plain-shift yaws, BORDER_CONSTANT 0, view rows of whole dwords).
//
// Rot column r of a yaw with shift s blends source columns (r + s) mod pw and the next one with the yaw's weight f
// (YawDesc), except the one column whose left source is pw - 1: P:105 clips it there, it is a copy.  A pixel's two
// rot taps r, r + 1 therefore read THREE contiguous source pixels -- also across the row's end, where the device rows
// carry a copy of their first pixels (PANO_PAD): one 12-byte load per row.  Per (tile, yaw) one of three loops:
//   0  whole-column shift and no pixel of the tile at the seam: a, b are the first two pixels of the load (copy)
//   1  blend, no pixel at the seam: the shift (s, or s - pw when every pixel is past the seam) is one constant per pair
//   2  the seam or the clipped column inside the tile's columns (always so next to a pole): wrap test per pixel,
//      the two rot taps patched to copies where their left source is pw - 1
// Per pixel and pair: 2 aligned 12-byte loads; 3 re-alignments and 12 blend instructions per row (class 1) + the 14 of
// stage 2 -- about 50 instructions whatever the footprint, against 16 + 7 per rot pixel of the footprint in the LDS scheme.
// ---------------------------------------------------------------------------------------------
// SINGLE: one set of tap registers (a pair's taps are loaded, waited for, blended: no prefetch inside the wave) -- 28
// registers less, for the band kernel's merged launch, whose other workgroups must keep their six waves per SIMD.
template <bool SINGLE = false>
__device__ __forceinline__ void draw_gather(
    const ViewsParams& P, const uint8_t* __restrict__ src, const YawDesc* __restrict__ ydesc, uint8_t* __restrict__ out,
    const TileGeo& G, uint32_t* stage, int chunk = -1)
{
    constexpr int PXT = VIEWS_PXT;
    const int t = threadIdx.x;
    // the pixels' quantised coordinates first: their address follows from the tile's position alone, the loads fly
    // while the header and the yaw descriptors arrive
    // Which pixels a wave instruction draws.  Rows: lane l draws column l of row j of the wave's four rows -- one or two
    // cache lines per load where an output row runs along a source row.  Blocks (G.blocky, set by the plan where the
    // output rows run ACROSS the source rows, next to a pole: 64 lanes, 64 lines, and the kernel waits for the L1's one
    // line per cycle): lane l draws pixel (l & 15, l >> 4) of the j-th 16 x 4 block of the wave's 64 x 4 strip.
    // Either way a lane's four stored pixels are four neighbours in one row (store_ctx).
    constexpr int WPR = TILE_W / 64;  // waves side by side in a tile
    const int wv = t >> 6, ln = t & 63;
    int pxs[PXT], pys[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        pxs[j] = G.x0 + (G.blocky ? 64 * (wv % WPR) + 16 * j + (ln & 15) : G.col);
        pys[j] = G.y0 + (G.blocky ? 4 * (wv / WPR) + (ln >> 4) : G.row0 + j * TILE_ROWSTEP);
    }
    int2 cxy[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        cxy[j] = make_int2(INT32_MIN, INT32_MIN);
        if (pxs[j] < P.ow && pys[j] < P.oh)
            cxy[j] = P.coords[((size_t)G.pitch_i * P.oh + pys[j]) * P.ow + pxs[j]];
    }
    // the rot columns this tile taps, as the plan's header has them -- validated: everything below stays inside a
    // panorama for any header
    const int last_col = P.pw - 1;
    // (a live pixel's left tap may sit one column outside: -1 is a legal c0)
    P2P_AUD_LT(P.audit, AUD_GATHER_BOX, G.c1 < G.c0 ? 0 : G.c0 + 1, P.pw + 1);
    P2P_AUD_LT(P.audit, AUD_GATHER_BOX, G.c1 < G.c0 ? 0 : G.c1 + 1, P.pw + 1);
    const int c0v = G.c0 < 0 ? 0 : (G.c0 > last_col ? last_col : G.c0);
    const int c1v = G.c1 < c0v ? c0v : (G.c1 > last_col ? last_col : G.c1);

    // ---- pair contexts (lane k), sorted by class ----
    int pair0, npairs;
    pair_chunk(P, false, P.gather_ppb, chunk >= 0 ? chunk : list_grid_chunk(), pair0, npairs);
    int cwA = 0, cwB = 0, cwC = 0, cw3 = 0, cls = 3;
    const int k = t & 63;
    if (k < npairs) {
        const int pair = pair0 + k;
        cw3 = pano_of_pair(P, pair);
        const int yi = pair - cw3 * P.n_yaw;
        const YawDesc yd = ydesc[yi];
        cw3 |= k << 26;
        if (!view_wanted(P, G.pitch_i, yi)) {
            cls = 4;  // a view the job does not draw (p2p_job_set_view_mask)
        } else if (yd.mode == 0 && (unsigned)yd.s < (unsigned)P.pw && (unsigned)yd.f <= 32u) {
            const bool nowrap = c1v + 1 + yd.s <= P.pw - 2, allwrap = c0v + yd.s >= P.pw;
            // (a whole-column shift has no clipped column; the seam itself still needs the wrap test)
            cls = (nowrap || allwrap) ? (yd.f == 0 ? 0 : 1) : 2;
            cwA = 3 * (allwrap ? yd.s - P.pw : yd.s);
            cwB = yd.f | (yd.c_clamp + 1) << 8;
            cwC = P.pw - yd.s;  // rot columns from here on read past the row's end
        }
    }
    int cum[5];
    const int r = sort_lanes_by_class<5>(k, k < npairs, cls, cum);
    if (cum[2] == 0)
        return;  // every yaw of the chunk is the table kernel's, or not wanted
    cwA = __builtin_amdgcn_ds_permute(4 * r, cwA);
    cwB = __builtin_amdgcn_ds_permute(4 * r, cwB);
    cwC = __builtin_amdgcn_ds_permute(4 * r, cwC);
    cw3 = __builtin_amdgcn_ds_permute(4 * r, cw3);

    // ---- this thread's pixels: tap weights, the two row offsets of its left rot tap ----
    uint32_t off_up[PXT], d_lo[PXT];  // byte offset of the upper row's left source pixel at shift 0; lower row - upper row
    int rx[PXT];
    TapWeights tw[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const bool inside = pxs[j] < P.ow && pys[j] < P.oh;
        const int2 c = cxy[j];
        const int ix = sat_short(c.x >> 5), iy = sat_short(c.y >> 5);
        // cv::remap, BORDER_CONSTANT 0: a pixel whose 2x2 footprint misses the panorama is black, a tap outside it
        // reads 0 -- weight 0 here, and an address that stays inside
        const bool live = inside && ix >= -1 && iy >= -1 && ix < P.pw && iy < P.ph;
        TapWeights w = tap_weights((uint32_t)c.x & 31u, (uint32_t)c.y & 31u, live);
        int x = ix;
        if (ix + 1 > last_col) {  // right taps outside
            w.w_up &= 0xFFFFu;
            w.w_lo &= 0xFFFFu;
        }
        if (ix < 0) {             // left taps outside: the right tap (rot column 0) moves to the left slot
            w.w_up >>= 16;
            w.w_lo >>= 16;
            x = 0;
        }
        if (iy < 0)
            w.w_up = 0u;
        if (iy + 1 > P.ph - 1)
            w.w_lo = 0u;
        tw[j] = w;
#ifdef P2P_AUDIT
        if (live)
            P2P_AUD_LT(P.audit, AUD_GATHER_COORD, x - c0v, c1v - c0v + 1);
#endif
        x = (!live || x < c0v) ? c0v : (x > c1v ? c1v : x);
        const int yu = iy < 0 ? 0 : (iy > P.ph - 1 ? P.ph - 1 : iy);
        const int yl = iy + 1 < 0 ? 0 : (iy + 1 > P.ph - 1 ? P.ph - 1 : iy + 1);
        rx[j] = x;
        off_up[j] = (uint32_t)yu * (uint32_t)P.src_pitch + 3u * (uint32_t)x;
        d_lo[j] = (uint32_t)(yl - yu) * (uint32_t)P.src_pitch;
    }
    const size_t view_bytes = P.view_bytes;
    StoreCtx SC = store_ctx(P, G, stage, t);
    if (G.blocky) {  // lanes x4 .. x4 + 3 of instruction sj: columns 16 sj + (x4 & 15) ... of row x4 >> 4 of the strip
        const int x4 = 4 * (ln & 15), sj = ln >> 4;
        const int srow = 4 * (wv / WPR) + (x4 >> 4), scol = 64 * (wv % WPR) + 16 * sj + (x4 & 15);
        const bool s_ok = srow < TILE_H && G.y0 + srow < P.oh && G.x0 + scol < P.ow;
        SC.out_off12 = s_ok ? (uint32_t)(G.y0 + srow) * (uint32_t)P.out_row + 3u * (uint32_t)(G.x0 + scol) : 0xFFFFFFFFu;
    }
    const uint32_t row_bytes = 3u * (uint32_t)P.pw;
    uint32_t bias_br = 0x00800080u;
    asm volatile("" : "+v"(bias_br));

    // Loads are 12 bytes from a 4-byte-aligned address (a vector load from an odd address is served byte by byte:
    // 240 cycles per wave instead of 12); the three pixels start at byte 0..3 of the load, v_alignbyte_b32 brings
    // them to byte 0.  The rows of a panorama are 16 bytes apart modulo 16, so both rows share the alignment.
    struct TapRegs {
        bu32x3 U[PXT], L[PXT];
        uint32_t al[PXT];
    };
    TapRegs ta, tb;  // the loads of pair k + 1 are in flight while pair k is blended: two sets, used alternately
    // MODE as the classes above
    auto load_taps = [&](auto mode_c, int kk, TapRegs& R) {
        constexpr int MODE = decltype(mode_c)::value;
        const auto S = make_buf(src + (size_t)(__builtin_amdgcn_readlane(cw3, kk) & 0x3FFFFFF) * P.pano_stride, (uint32_t)P.pano_stride);
        const uint32_t shift3 = (uint32_t)__builtin_amdgcn_readlane(cwA, kk);
        const int seam = __builtin_amdgcn_readlane(cwC, kk);
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            uint32_t a = off_up[j] + shift3;
            if (MODE == 2)
                a = rx[j] >= seam ? a - row_bytes : a;
            R.al[j] = a;  // its two low bits are the alignment
            a &= ~3u;
            P2P_AUD_RANGE(P.audit, AUD_GATHER_SRC, a + d_lo[j], 12u, P.pano_stride);
            R.U[j] = __builtin_amdgcn_raw_buffer_load_b96(S, (int)a, 0, P2P_GATHER_LOAD_AUX);
            R.L[j] = __builtin_amdgcn_raw_buffer_load_b96(S, (int)(a + d_lo[j]), 0, P2P_GATHER_LOAD_AUX);
        }
    };
    // the two rot taps of one row from its three source pixels (bytes o .. o + 8 of the load)
    auto rot_pair = [&](auto mode_c, const bu32x3& q, uint32_t o, uint32_t f8, uint32_t g8, bool a_clip, bool b_clip, uint32_t& a, uint32_t& b) {
        constexpr int MODE = decltype(mode_c)::value;
        const uint32_t u0 = __builtin_amdgcn_alignbyte(q.y, q.x, o), u1 = __builtin_amdgcn_alignbyte(q.z, q.y, o);
        const uint32_t a_cp = u0 & 0x00FFFFFFu, b_cp = __builtin_amdgcn_perm(u1, u0, 0x0C050403u);
        if (MODE == 0) {
            a = a_cp;
            b = b_cp;
            return;
        }
        const uint32_t u2 = __builtin_amdgcn_alignbyte(q.z, q.z, o);        // byte 8 of the run in its byte 0
        const uint32_t m0 = u0 & 0x00FF00FFu;                               // B0 R0  (bytes 0, 2)
        const uint32_t m1 = __builtin_amdgcn_perm(u1, u0, 0x0C050C03u);     // B1 R1  (3, 5)
        const uint32_t m2 = __builtin_amdgcn_perm(u2, u1, 0x0C040C02u);     // B2 R2  (6, 8)
        const uint32_t n01 = __builtin_amdgcn_perm(u1, u0, 0x0C040C01u);    // G0 G1  (1, 4)
        const uint32_t n12 = __builtin_amdgcn_perm(u1, u1, 0x0C030C00u);    // G1 G2  (4, 7)
        const uint32_t br0 = vmad24(f8, m1, vmad24(g8, m0, bias_br));
        const uint32_t br1 = vmad24(f8, m2, vmad24(g8, m1, bias_br));
        const uint32_t g01 = vmad24(f8, n12, vmad24(g8, n01, bias_br));
        a = __builtin_amdgcn_perm(br0, g01, 0x0C070105u);
        b = __builtin_amdgcn_perm(br1, g01, 0x0C070305u);
        if (MODE == 2) {
            a = a_clip ? a_cp : a;
            b = b_clip ? b_cp : b;
        }
    };
    auto one_pair = [&](auto mode_c, int kk, int kend, const TapRegs& cur, TapRegs& nxt) {
        // the next pair's taps go out first (the last pair asks for its own again: no branch on the memory path);
        // "this pair's taps landed" is then a counted vmcnt: the previous pair's store and these loads may be pending
        load_taps(mode_c, kk + 1 < kend ? kk + 1 : kk, nxt);
        const uint32_t wB = (uint32_t)__builtin_amdgcn_readlane(cwB, kk);
        const uint32_t f8 = 8u * (wB & 0xFFu), g8 = 256u - f8;
        const int cc = (int)(wB >> 8) - 1;  // the rot column clipped to source column pw - 1, or -1
        uint32_t pix[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const bool a_clip = rx[j] == cc, b_clip = rx[j] + 1 == cc;
            uint32_t a, b, c, d;
            rot_pair(mode_c, cur.U[j], cur.al[j], f8, g8, a_clip, b_clip, a, b);
            rot_pair(mode_c, cur.L[j], cur.al[j], f8, g8, a_clip, b_clip, c, d);
            pix[j] = blend4_packed(a, b, c, d, tw[j]);
        }
        const int pair = pair0 + (int)((uint32_t)__builtin_amdgcn_readlane(cw3, kk) >> 26);
        store_wave_pixels(SC, pix, out + ((size_t)pair * P.n_pitch + G.pitch_i) * view_bytes, view_bytes);
    };
    auto run = [&](auto mode_c, int kbeg, int kend) {
        if (kbeg >= kend)
            return;
        if (SINGLE) {
            for (int kk = kbeg; kk < kend; ++kk) {
                load_taps(mode_c, kk, ta);
                const uint32_t wB = (uint32_t)__builtin_amdgcn_readlane(cwB, kk);
                const uint32_t f8 = 8u * (wB & 0xFFu), g8 = 256u - f8;
                const int cc = (int)(wB >> 8) - 1;
                uint32_t pix[PXT];
#pragma unroll
                for (int j = 0; j < PXT; ++j) {
                    const bool a_clip = rx[j] == cc, b_clip = rx[j] + 1 == cc;
                    uint32_t a, b, c, d;
                    rot_pair(mode_c, ta.U[j], ta.al[j], f8, g8, a_clip, b_clip, a, b);
                    rot_pair(mode_c, ta.L[j], ta.al[j], f8, g8, a_clip, b_clip, c, d);
                    pix[j] = blend4_packed(a, b, c, d, tw[j]);
                }
                const int pair = pair0 + (int)((uint32_t)__builtin_amdgcn_readlane(cw3, kk) >> 26);
                store_wave_pixels(SC, pix, out + ((size_t)pair * P.n_pitch + G.pitch_i) * view_bytes, view_bytes);
            }
            return;
        }
        load_taps(mode_c, kbeg, ta);
        // one store that writes nothing: the loop is entered with the vmcnt shape it has inside (see draw_tight)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_buffer_store_b32(0u, __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000), 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        int kk = kbeg;
        for (; kk + 1 < kend; kk += 2) {
            one_pair(mode_c, kk, kend, ta, tb);
            one_pair(mode_c, kk + 1, kend, tb, ta);
        }
        if (kk < kend)
            one_pair(mode_c, kk, kend, ta, tb);
    };
    run(std::integral_constant<int, 0>{}, 0, cum[0]);
    run(std::integral_constant<int, 1>{}, cum[0], cum[1]);
    run(std::integral_constant<int, 2>{}, cum[1], cum[2]);
}

// ---------------------------------------------------------------------------------------------
// Rest / table kernel body: same arithmetic, every case distinction.
// ---------------------------------------------------------------------------------------------
struct PairCtx {      // uniform per (tile, pair)
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

// table path: (3*i | f << 20) table entry, 8-byte load of pixels i and i+1 through the panorama's descriptor
__device__ __forceinline__ uint32_t rot_pixel_buf(const ViewsParams& P, __amdgpu_buffer_rsrc_t S, uint32_t row_off, uint32_t te, uint32_t site)
{
    const uint32_t off = row_off + (te & 0xFFFFFu);
    P2P_AUD_RANGE(P.audit, site, off, 8u, P.pano_stride);
    const bu32x2 q = __builtin_amdgcn_raw_buffer_load_b64(S, (int)off, 0, 0);
    const uint32_t f = te >> 20;
    return rot_blend2(q.x, __builtin_amdgcn_alignbyte(q.y, q.x, 3), f, 32u - f);
}

// TABLE = true: the body of remap_views_table_kernel -- a mode 2 tile, a few pairs per workgroup; false:
// remap_views_rest_kernel -- the general loop over the mode 1 tiles.
// BORDER (table kernel only): the job's border mode, a compile-time constant -- 0 = BORDER_CONSTANT (the current tool,
// P:192-199) with its tap masks, 1..4 = the other cv2 codes with THAT mode's cv::borderInterpolate arithmetic (the legacy
// tool, L:179: BORDER_REFLECT): the mode is the LAUNCH's, not the pixel's, and an instance carries one mode's code.
template <bool TABLE, int BORDER = 0>
__device__ __forceinline__ void draw_rest(
    const ViewsParams& P, const uint8_t* __restrict__ src, const uint32_t* __restrict__ ytab,
    const YawDesc* __restrict__ ydesc, const uint32_t* __restrict__ f4tab, uint8_t* __restrict__ out,
    const TileGeo& G, const uint32_t* __restrict__ pxw, const uint32_t* __restrict__ itw,
    uint4 (*tile4)[LDS_ITEMS_CAP])
{
    constexpr int PXT = VIEWS_PXT;
    const int t = threadIdx.x;
    const bool main_draws_plain = tight_tile(G, P);
    const int px = G.x0 + G.col, py0 = G.y0 + G.row0;
    bool inside[PXT], inside4[PXT];  // the pixel is in the view; its 4-pixel group starts in the view
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const bool row_ok = G.row0 + j * TILE_ROWSTEP < TILE_H && py0 + j * TILE_ROWSTEP < P.oh;
        inside[j] = row_ok && px < P.ow;
        inside4[j] = row_ok && (px & ~3) < P.ow;
    }

    // output addressing: 4 horizontally adjacent pixels = 12 bytes = 3 aligned dwords
    const int lane4 = t & 3;
    const size_t view_bytes = P.view_bytes;
    const uint32_t pix_off = (uint32_t)py0 * (uint32_t)P.out_row + 3u * (uint32_t)px;  // < 3 * 32768^2 < 2^32
    const uint32_t pix_step = (uint32_t)TILE_ROWSTEP * (uint32_t)P.out_row;
    // dword lane4 of the 12 bytes P0 P1 P2 P3: bytes of the own pixel (0-2) and of the next lane's (4-6)
    const uint32_t store_sel = lane4 == 0 ? 0x04020100u : (lane4 == 1 ? 0x05040201u : 0x06050402u);

    auto store_pixels = [&](int pair, const uint32_t (&pix)[PXT]) {
        uint8_t* O = out + ((size_t)pair * P.n_pitch + G.pitch_i) * view_bytes;  // [pano][yaw][pitch][oh][ow][3]
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const uint32_t off = pix_off + (uint32_t)j * pix_step;
            // lanes 4k..4k+3 hold pixels P0..P3; lanes with lane4 < 3 emit dword lane4 of the 12 bytes
            // neighbour lane's pixel: row_shl:1 DPP (lane4 groups never straddle a 16-lane row)
            uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pix[j], 0x101, 0xF, 0xF, true);
            uint32_t dw = __builtin_amdgcn_perm(nxt, pix[j], store_sel);
            uint32_t voff = off + (uint32_t)lane4;
            asm volatile("" : "+v"(voff));
            // (a group that starts inside the view may end in the row's padding: the device row holds whole groups)
            if (inside4[j] && lane4 < 3)
                __builtin_nontemporal_store(dw, reinterpret_cast<uint32_t*>(O + voff));
        }
    };

    // ---- table path: the quantised coordinates come from the plan's coordinate dump ----
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
                c = P.coords[((size_t)G.pitch_i * P.oh + (py0 + j * TILE_ROWSTEP)) * P.ow + px];
            d.ix[j] = sat_short(c.x >> 5);
            d.iy[j] = sat_short(c.y >> 5);
            d.fx[j] = (uint32_t)c.x & 31u;
            d.fy[j] = (uint32_t)c.y & 31u;
            // A pixel contributes only if its 2x2 footprint touches the panorama (BORDER_CONSTANT 0: cv::remap
            // writes borderValue when sx >= w || sx+1 < 0 || sy >= h || sy+1 < 0)
            const bool inrange = inside[j] && d.ix[j] >= -1 && d.iy[j] >= -1 && d.ix[j] < P.pw && d.iy[j] < P.ph;
            // other border modes (legacy entry point, L:179) resolve every tap to some pixel
            d.live[j] = BORDER == 0 ? inrange : inside[j];
        }
    };
    auto direct_pixels = [&](const DirectPx& d, __amdgpu_buffer_rsrc_t S, int yi, uint32_t (&pix)[PXT]) {
        // same arithmetic, taps gathered from global memory through the packed yaw table; the table's entries address
        // the panorama through its descriptor
        const uint32_t* __restrict__ T = ytab + (size_t)yi * P.pw;
        const uint32_t pitch = (uint32_t)P.src_pitch;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            pix[j] = 0;
            if (BORDER != 0) {
                if (!d.live[j])
                    continue;
                const int xa = border_interpolate(d.ix[j], P.pw, BORDER), xb = border_interpolate(d.ix[j] + 1, P.pw, BORDER);
                const int ya = border_interpolate(d.iy[j], P.ph, BORDER), yb = border_interpolate(d.iy[j] + 1, P.ph, BORDER);
                const uint32_t row0 = (uint32_t)ya * pitch, row1 = (uint32_t)yb * pitch;
                const uint32_t t0 = T[xa], t1 = T[xb];
                pix[j] = blend4(rot_pixel_buf(P, S, row0, t0, AUD_TABLE_SRC), rot_pixel_buf(P, S, row0, t1, AUD_TABLE_SRC),
                                rot_pixel_buf(P, S, row1, t0, AUD_TABLE_SRC), rot_pixel_buf(P, S, row1, t1, AUD_TABLE_SRC),
                                d.fx[j], d.fy[j]);
                // (one pixel's taps at a time: four pixels' worth of border arithmetic and loads in flight cost 140 registers)
                __builtin_amdgcn_sched_barrier(0);
            } else if (d.live[j]) {
                const bool c0in = d.ix[j] >= 0, c1in = d.ix[j] + 1 < P.pw, r0in = d.iy[j] >= 0, r1in = d.iy[j] + 1 < P.ph;
                const uint32_t row0 = (uint32_t)(r0in ? d.iy[j] : 0) * pitch, row1 = (uint32_t)(r1in ? d.iy[j] + 1 : 0) * pitch;
                const uint32_t t0 = c0in ? T[d.ix[j]] : 0u, t1 = c1in ? T[d.ix[j] + 1] : 0u;
                uint32_t a = (c0in && r0in) ? rot_pixel_buf(P, S, row0, t0, AUD_TABLE_SRC) : 0u;
                uint32_t b = (c1in && r0in) ? rot_pixel_buf(P, S, row0, t1, AUD_TABLE_SRC) : 0u;
                uint32_t c = (c0in && r1in) ? rot_pixel_buf(P, S, row1, t0, AUD_TABLE_SRC) : 0u;
                uint32_t dd = (c1in && r1in) ? rot_pixel_buf(P, S, row1, t1, AUD_TABLE_SRC) : 0u;
                pix[j] = blend4(a, b, c, dd, d.fx[j], d.fy[j]);
            }
        }
    };

    if constexpr (TABLE) {
        if (G.mode == 1)
            return;
        DirectPx d;
        load_direct(d);
        const bool list = P.use_pair_list != 0;
        int first, count;
        pair_chunk(P, list, P.gather_ppb, list_grid_chunk(), first, count);
        for (int k = 0; k < count; ++k) {
            const int pair = pair_of_lane(P, list, first, k, AUD_TABLE_PAIR);
            const int pano_i = pano_of_pair(P, pair);
            const int yaw_i = pair - pano_i * P.n_yaw;
            if (!view_wanted(P, G.pitch_i, yaw_i))
                continue;  // (uniform: the pair is the workgroup's)
            uint32_t pix[PXT];
            direct_pixels(d, make_buf(src + (size_t)pano_i * P.pano_stride, (uint32_t)P.pano_stride), yaw_i, pix);
            store_pixels(pair, pix);
        }
        return;
    }
    if (G.mode != 1)
        return;  // the gather / table kernels'
    P2P_AUD_LT(P.audit, AUD_REST_HDR, G.n_items, LDS_ITEMS_CAP + 1);

    // ---- LDS scheme, general loop ----
    const bool listed = P.use_pair_list != 0;
    const PairCtxs X = listed ? pair_contexts<true>(P, ydesc, G.c0, G.c1, t, tile_grid_chunk(P), G.pitch_i)
                              : pair_contexts<false>(P, ydesc, G.c0, G.c1, t, tile_grid_chunk(P), G.pitch_i);
    const int kfirst = main_draws_plain ? X.n3 : 0;  // the main kernel has classes 0..3 of its tiles
    const int klast = X.n4;                          // (class 5: views the job does not draw)
    if (kfirst >= klast)
        return;
    uint32_t tap_up[PXT], tap_lo[PXT];
    TapWeights tw[PXT];
    decode_px<PXT>(pxw, t, tap_up, tap_lo, tw);
    uint32_t slot_off[VIEWS_SLOTS], slot_g[VIEWS_SLOTS];
    decode_items(itw, t, G.n_items, P.src_pitch, slot_off, slot_g);
    const int wave_base = __builtin_amdgcn_readfirstlane(t & ~63);
    const int n_items = G.n_items < LDS_ITEMS_CAP ? G.n_items : LDS_ITEMS_CAP;
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
        P2P_AUD_LT(P.audit, AUD_REST_PAIR, c.yaw_i, P.n_yaw);
        c.yaw_i = c.yaw_i < P.n_yaw ? c.yaw_i : P.n_yaw - 1;  // it indexes the yaw tables and the output views
        c.cf0 = __builtin_amdgcn_readlane(X.cw2, k);
        const int w3 = __builtin_amdgcn_readlane(X.cw3, k);
        c.pano = w3 & 0x3FFFFFF;
        c.korig = (int)((uint32_t)w3 >> 26);
        return c;
    };
    auto issue_loads = [&](const PairCtx& pc, __amdgpu_buffer_rsrc_t S, Q16 (&q)[VIEWS_SLOTS],
                           uint32_t (&fw)[VIEWS_SLOTS]) {
#pragma unroll
        for (int k = 0; k < VIEWS_SLOTS; ++k) {
            if (wave_base + k * VIEWS_BLOCK < n_items) {  // a wave runs slot k only if its first lane has an item there
                uint32_t off = slot_off[k] + pc.goff;
                if (slot_g[k] >= pc.wrap_g)
                    off -= row_bytes;
                P2P_AUD_RANGE(P.audit, AUD_REST_SRC, off, 16u, P.pano_stride);
                const bu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(S, (int)off, 0, 0);
                q[k].d[0] = v.x; q[k].d[1] = v.y; q[k].d[2] = v.z; q[k].d[3] = v.w;
                if (pc.per_column) {
                    // rot column of the item's first pixel: its source column - s (mod pw)
                    int cf = 4 * (int)slot_g[k] + pc.cf0;
                    if (slot_g[k] >= pc.wrap_g)
                        cf -= P.pw;
                    if (cf < 0)
                        cf += P.pw;
                    P2P_AUD_LT(P.audit, AUD_REST_F4, cf, P.pw);
                    cf = (unsigned)cf < (unsigned)P.pw ? cf : 0;
                    fw[k] = f4tab[(size_t)pc.yaw_i * P.pw + cf];
                }
            }
        }
    };

    PairCtx pc = pair_ctx(kfirst);
    Q16 q[VIEWS_SLOTS];
    uint32_t fw[VIEWS_SLOTS];
    if (pc.fast)
        issue_loads(pc, make_buf(src + (size_t)pc.pano * P.pano_stride, (uint32_t)P.pano_stride), q, fw);
    int buf = 0;
    for (int ki = kfirst; ki < klast; ++ki) {
        const auto S = make_buf(src + (size_t)pc.pano * P.pano_stride, (uint32_t)P.pano_stride);
        const int cur_yaw = pc.yaw_i;
        const int pair = listed ? pc.pano * P.n_yaw + pc.yaw_i : X.pair0 + pc.korig;
        const bool has_next = ki + 1 < klast;
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
                    issue_loads(pc, make_buf(src + (size_t)pc.pano * P.pano_stride, (uint32_t)P.pano_stride), q, fw);
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
                    issue_loads(pc, make_buf(src + (size_t)pc.pano * P.pano_stride, (uint32_t)P.pano_stride), q, fw);
            }
        }
        store_pixels(pair, pix);
    }
}

// the plan's list of mode 2 tiles -> (pitch, tile); an entry beyond the plan's slots (never written by the plan
// pass) is clamped
// XCD_LISTS (the gather kernel): the list is [8][P.n_list], one work list per XCD, grouped by the tiles' position in
// the SOURCE so that tiles of different pitch views that read the same part of the panorama meet in one L2
// (p2p_host_plan.cpp: xcd_lists); gridDim.x == 8 * n_list, ~0 = no tile (returns false).
template <bool XCD_LISTS>
__device__ __forceinline__ bool tile_of_list(const ViewsParams& P, const uint32_t* __restrict__ list, uint32_t site, int& pitch_i, int& tile_id,
                                             uint32_t bx = blockIdx.x, int n_list = -1)
{
    const uint32_t tiles = (uint32_t)(((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H));
    uint32_t idx = bx;
    if (XCD_LISTS)
        idx = (bx & 7u) * (uint32_t)(n_list >= 0 ? n_list : P.n_list) + (bx >> 3);
    uint32_t slot = list[idx];
    if (XCD_LISTS && slot == ~0u)
        return false;
    P2P_AUD_LT(P.audit, site, slot, tiles * (uint32_t)P.n_pitch);
    if (slot >= tiles * (uint32_t)P.n_pitch)
        slot = 0u;
    pitch_i = (int)(slot / tiles);
    tile_id = (int)(slot - (uint32_t)pitch_i * tiles);
    return true;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// The job's pair-context table: one wave per (tile slot, chunk of pairs) runs pair_contexts and stores what the view
// kernels' workgroups would each work out again (pair_contexts_from_table).  band != 0: the slots are band tiles.
__global__ __launch_bounds__(64) void pair_ctx_kernel(ViewsParams P, uint4* __restrict__ table, int band)
{
    const uint32_t slot = blockIdx.x;
    const int chunk = (int)blockIdx.y, t = threadIdx.x;
    int c0, c1, pitch_i = -1;
    uint32_t mode;
    if (band) {
        const PieceHdr h = P.band_hdr[slot];
        c0 = h.c0; c1 = h.c1; mode = 1u;
    } else {
        const uint32_t tiles = (uint32_t)(((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H));
        const PieceHdr h = P.hdr[slot];
        c0 = h.c0; c1 = h.c1; mode = h.mode_items & 3u;
        pitch_i = (int)(slot / tiles);
    }
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (mode == 1u) {
        const PairCtxs X = pair_contexts(P, P.ydesc, c0, c1, t, chunk, pitch_i, 0);
        r.x = X.cw0; r.y = X.cw1; r.z = (uint32_t)X.cw3;
        r.w = t == 0 ? ((uint32_t)X.n0 | (uint32_t)X.n1 << 8 | (uint32_t)X.n2 << 16 | (uint32_t)X.n3 << 24)
                     : (t == 1 ? ((uint32_t)X.n4 | (uint32_t)X.npairs << 8) : 0u);
    }
    table[((size_t)slot * gridDim.y + (size_t)chunk) * 64u + (size_t)t] = r;
}

hipError_t launch_pair_ctx(const ViewsParams& P, uint4* table, int slots, int chunks, int band, hipStream_t st)
{
    if (slots <= 0 || chunks <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(pair_ctx_kernel, dim3(slots, chunks), dim3(64), 0, st, P, table, band);
    return hipGetLastError();
}

// MERGED (list order only: a one-dimensional grid): the launch's first 8 * merge_gather_n * (chunks of gather_ppb pairs)
// workgroups draw the plan's gather tiles -- the gather kernel's body with ONE set of tap registers, so that this kernel
// keeps its registers -- instead of a launch of their own in front of this one: they are few, long chains of memory
// latencies (the tiles around a pole), and hide behind the LDS-scheme tiles.
template <bool MERGED>
__global__ __launch_bounds__(VIEWS_BLOCK, VIEWS_WAVES_PER_SIMD) void remap_views_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, const YawDesc* __restrict__ ydesc, uint8_t* __restrict__ out,
    const PieceHdr* __restrict__ hdr, const uint32_t* __restrict__ px, const uint32_t* __restrict__ items,
    const uint32_t* __restrict__ main_count)
{
    __shared__ uint4 tile4[2][LDS_ITEMS_CAP];
    uint32_t bx = blockIdx.x;
    if (MERGED) {
        const uint32_t per_chunk = 8u * (uint32_t)P.merge_gather_n;
        const uint32_t n_pairs_all = (uint32_t)(P.n_panos * P.n_yaw);
        const uint32_t gw = per_chunk * ((n_pairs_all + (uint32_t)P.gather_ppb - 1u) / (uint32_t)P.gather_ppb);
        if (bx < gw) {
            const uint32_t gchunk = bx / per_chunk;
            int gp, gt;
            if (!tile_of_list<true>(P, P.merge_gather_list, AUD_GATHER_LIST, gp, gt, bx - gchunk * per_chunk, P.merge_gather_n))
                return;
            const int gtiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
            const PieceHdr gh = hdr[(size_t)gp * gtiles + gt];
            const TileGeo GG = tile_geo(P, gh, gp, gt, (int)threadIdx.x);
            if (GG.mode != 2)
                return;
            static_assert(sizeof(tile4) >= (VIEWS_BLOCK / 64) * VIEWS_PXT * 64 * sizeof(uint32_t), "the gather body's staging dwords");
            draw_gather<true>(P, src, ydesc, out, GG, reinterpret_cast<uint32_t*>(&tile4[0][0]), (int)gchunk);
            return;
        }
        bx -= gw;
    }
#if defined(P2P_STAGE_OWN_LDS) || defined(P2P_STORE_INLINE)
    __shared__ __attribute__((aligned(16))) uint32_t stage[(VIEWS_BLOCK / 64) * VIEWS_PXT * 64];  // a dword per pixel
#else
    uint32_t* const stage = nullptr;  // staged inside the tile buffers (draw_tight)
#endif
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    int tile_id, pitch_i, chunk, ppb = 0;
    // (64-wide shapes only: the 128-wide kernel is the shape of launches of several GB, which draw several chunks of pairs
    // per tile -- no split tail there -- and it sits at its 80 registers)
    if (TILE_W != 128 && P.main_list && P.main_tail > 0) {
        // List order, ONE chunk of pairs, no prefetch workgroups (the host's rule): entry q of the XCD's list -- but its
        // last main_tail entries are drawn by main_tail_parts workgroups each, a part of the pairs each (p2p_host_job.cpp: main_tail).
        // (the XCD's count: written with the list by main_lists_kernel -- a scalar load through the restrict-qualified parameter)
        const uint32_t q = bx >> 3, L = min(main_count[bx & 7u], (uint32_t)P.main_stride);
        const uint32_t K = (uint32_t)P.main_tail < L ? (uint32_t)P.main_tail : L;
        uint32_t e = q;
        chunk = 0;
        if (q >= L - K) {
            const uint32_t i = q - (L - K), parts = (uint32_t)P.main_tail_parts;
            if (i >= parts * K)
                return;
            e = L - K + i / parts;
            chunk = (int)(i - (i / parts) * parts);
            ppb = (P.n_panos * P.n_yaw + (int)parts - 1) / (int)parts;
        }
        if (e >= (uint32_t)P.main_stride)
            return;
        uint32_t slot = P.main_list[(bx & 7u) * (uint32_t)P.main_stride + e];
        if (slot == ~0u)
            return;
        P2P_AUD_LT(P.audit, AUD_MAIN_PITCH, slot, (uint32_t)(tiles * P.n_pitch));
        slot = slot < (uint32_t)(tiles * P.n_pitch) ? slot : 0u;  // (a garbage list draws a valid tile)
        pitch_i = (int)(slot / (uint32_t)tiles);
        tile_id = (int)(slot - (uint32_t)pitch_i * (uint32_t)tiles);
    } else if (P.main_list) {
        // List order (p2p_host_plan.cpp: xcd_main_lists): workgroup b runs on XCD b & 7 and is that XCD's q-th, q = b >> 3.
        // The XCD draws main_group entries of its list for one chunk of pairs, the same entries for the next chunk
        // (their plan tables and source rows are still in its L2), and so on, then the next main_group entries.
        // With pf_lead > 0 (plan tables beyond the Infinity Cache: config 4) one more workgroup per block, dispatched
        // ahead of the block's last chunk, draws nothing and touches the plan tables of the NEXT block's entries.
        const uint32_t q = bx >> 3, group = (uint32_t)P.main_group, chunks = (uint32_t)P.main_chunks;
        const uint32_t pf = P.pf_lead > 0 ? 1u : 0u, pf_pos = group * (chunks - 1u);
        const uint32_t per_block = group * chunks + pf;
        const uint32_t blk = q / per_block;
        uint32_t r = q - blk * per_block;
        if (pf && r == pf_pos) {
            constexpr uint32_t LINES = PF_PX_LINES + PF_ITEM_LINES;
            const uint32_t* mine = P.main_list + (bx & 7u) * (uint32_t)P.main_stride;
            uint32_t acc = 0u;
            for (uint32_t i = threadIdx.x; i < group * LINES; i += VIEWS_BLOCK) {
                const uint32_t tj = i / LINES, l = i - tj * LINES, e = (blk + 1u) * group + tj;
                const uint32_t slot = e < (uint32_t)P.main_stride ? mine[e] : ~0u;
                if (slot < (uint32_t)(tiles * P.n_pitch)) {
                    const uint32_t* a = l < PF_PX_LINES ? px + (size_t)slot * (VIEWS_BLOCK * VIEWS_PXT) + l * 32
                                                        : items + (size_t)slot * LDS_ITEMS_CAP + (l - PF_PX_LINES) * 32;
                    acc |= *a;
                    if (l == 0)
                        acc |= hdr[slot].mode_items;
                }
            }
            asm volatile("" ::"v"(acc));
            return;
        }
        r -= (pf && r > pf_pos) ? 1u : 0u;
        chunk = (int)(r / group);
        const uint32_t e = blk * group + (r - (uint32_t)chunk * group);
        if (e >= (uint32_t)P.main_stride)
            return;
        uint32_t slot = P.main_list[(bx & 7u) * (uint32_t)P.main_stride + e];
        if (slot == ~0u)
            return;
        P2P_AUD_LT(P.audit, AUD_MAIN_PITCH, slot, (uint32_t)(tiles * P.n_pitch));
        slot = slot < (uint32_t)(tiles * P.n_pitch) ? slot : 0u;  // (a garbage list draws a valid tile)
        pitch_i = (int)(slot / (uint32_t)tiles);
        tile_id = (int)(slot - (uint32_t)pitch_i * (uint32_t)tiles);
    } else {
        const BlockRole role = main_block_role(P, (int)bx);
        // heaviest views first (the host orders pitch_order by |pitch - 90| descending): a smoother tail
        pitch_i = pitch_of_block(P, tile_grid_pitch_block(P));
        chunk = tile_grid_chunk(P);
        if (role.pf_count > 0) {
            // one dword of every 128-byte line of the tiles' tables; the values are not used (addresses follow from the
            // block index: inside the tables by construction)
            constexpr int LINES = PF_PX_LINES + PF_ITEM_LINES;
            uint32_t acc = 0u;
            const size_t slot0 = (size_t)pitch_i * tiles + role.pf_first;
            for (int i = (int)threadIdx.x; i < role.pf_count * LINES; i += VIEWS_BLOCK) {
                const int tj = i / LINES, l = i - tj * LINES;
                const uint32_t* a = l < PF_PX_LINES ? px + (slot0 + tj) * (VIEWS_BLOCK * VIEWS_PXT) + l * 32
                                                    : items + (slot0 + tj) * LDS_ITEMS_CAP + (l - PF_PX_LINES) * 32;
                acc |= *a;
            }
            if ((int)threadIdx.x < role.pf_count)
                acc |= hdr[slot0 + threadIdx.x].mode_items;
            asm volatile("" ::"v"(acc));
            return;
        }
        tile_id = role.tile_id;
        if (tile_id < 0)
            return;
    }
#ifdef P2P_ABLATE_ONE_TABLE  // timing experiment (right pixels only when all pitch views are the same): every pitch view reads the first one's plan tables
    const PieceHdr h = hdr[tile_id];
    TileGeo G = tile_geo(P, h, pitch_i, tile_id, (int)threadIdx.x);
    G.slot = (uint32_t)tile_id;
#elif defined(P2P_ABLATE_TABLE_WINDOW)  // timing experiment (wrong pixels): all plan-table reads inside a window of 64 tiles -- what tables of no size would give
    const PieceHdr h = hdr[(size_t)pitch_i * tiles + tile_id];
    TileGeo G = tile_geo(P, h, pitch_i, tile_id, (int)threadIdx.x);
    G.slot = (uint32_t)(tile_id & 63);
#else
    const PieceHdr h = hdr[(size_t)pitch_i * tiles + tile_id];
    const TileGeo G = tile_geo(P, h, pitch_i, tile_id, (int)threadIdx.x);
#endif
#ifdef P2P_ABLATE_ITEMS_WINDOW  // timing experiment (wrong pixels): the item lists of 64 tiles serve all -- what item lists of no size would give
    draw_tight(P, src, ydesc, out, G, px + (size_t)G.slot * (VIEWS_BLOCK * VIEWS_PXT), items + (size_t)(G.slot & 63u) * LDS_ITEMS_CAP,
               tile4, stage, chunk * (TILE_W == 128 && P.main_span > 1 ? P.main_span : 1), ppb);
#else
    draw_tight(P, src, ydesc, out, G, px + (size_t)G.slot * (VIEWS_BLOCK * VIEWS_PXT), items + (size_t)G.slot * LDS_ITEMS_CAP,
               tile4, stage, chunk * (TILE_W == 128 && P.main_span > 1 ? P.main_span : 1), ppb);
#endif
}

__global__ __launch_bounds__(VIEWS_BLOCK) void remap_views_rest_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, const uint32_t* __restrict__ ytab,
    const YawDesc* __restrict__ ydesc, const uint32_t* __restrict__ f4tab, uint8_t* __restrict__ out,
    const PieceHdr* __restrict__ hdr, const uint32_t* __restrict__ px, const uint32_t* __restrict__ items)
{
    __shared__ uint4 tile4[2][LDS_ITEMS_CAP];
    const int tile_id = tile_of_block(P, (int)blockIdx.x, (int)gridDim.x);
    if (tile_id < 0)
        return;
    const int pitch_i = pitch_of_block(P, tile_grid_pitch_block(P));
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const PieceHdr h = hdr[(size_t)pitch_i * tiles + tile_id];
    const TileGeo G = tile_geo(P, h, pitch_i, tile_id, (int)threadIdx.x);
    draw_rest<false>(P, src, ytab, ydesc, f4tab, out, G, px + (size_t)G.slot * (VIEWS_BLOCK * VIEWS_PXT),
                     items + (size_t)G.slot * LDS_ITEMS_CAP, tile4);
}

// One workgroup per (mode 2 tile of the plan's list, chunk of pairs).
#ifndef P2P_GATHER_WAVES
#define P2P_GATHER_WAVES 4  // two sets of tap registers: 4 waves per SIMD, 16 x 6 KB of loads in flight per CU
#endif
__global__ __launch_bounds__(VIEWS_BLOCK, P2P_GATHER_WAVES) void remap_views_gather_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, const YawDesc* __restrict__ ydesc, uint8_t* __restrict__ out,
    const PieceHdr* __restrict__ hdr, const uint32_t* __restrict__ gather_list)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage[(VIEWS_BLOCK / 64) * VIEWS_PXT * 64];
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    int pitch_i, tile_id;
    if (!tile_of_list<true>(P, gather_list, AUD_GATHER_LIST, pitch_i, tile_id))  // (gather_all: the list of ALL tiles)
        return;
    const PieceHdr h = hdr[(size_t)pitch_i * tiles + tile_id];
    const TileGeo G = tile_geo(P, h, pitch_i, tile_id, (int)threadIdx.x);
    if (G.mode != 2 && !P.gather_all)
        return;
    draw_gather(P, src, ydesc, out, G, stage);
}

// The band kernel: source-band tiles (p2p_device.h) for every plain-shift yaw.  The tiles are in source order (band by
// band); XCD x = blockIdx.x & 7 draws the run first[x] .. first[x + 1] - 1 of them (equal work, cut on the device:
// band_xcd_kernel), from its costlier end if the run says so, and the last band_tail tiles of the run by
// main_tail_parts workgroups each, a part of the pairs each (one chunk of pairs only: see main_tail).
#ifndef P2P_BAND_WAVES
#define P2P_BAND_WAVES 6  // (at 7 waves per SIMD, 72 registers, the three-item loops spill)
#endif
// MERGED: the launch's first 8 * band_gather_n * (chunks of gather_ppb pairs) workgroups draw the plan's gather tiles (the
// gather kernel's body with ONE set of tap registers, so that the kernel keeps its six waves per SIMD): few, long,
// latency-bound workgroups around a pole -- 17 us as a launch of their own behind which the band kernel waits.
template <bool MASKED, bool MERGED>
// (MASKED: the per-lane view word and its test cost 9-13 registers -- one wave per SIMD less instead of scratch)
__global__ __launch_bounds__(VIEWS_BLOCK, (MASKED && P2P_BAND_WAVES > 5) ? 5 : P2P_BAND_WAVES) void remap_views_band_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, const YawDesc* __restrict__ ydesc, uint8_t* __restrict__ out,
    const PieceHdr* __restrict__ hdr, const uint32_t* __restrict__ px, const uint32_t* __restrict__ grp, const BandInfo* __restrict__ info)
{
    __shared__ uint4 tile4[2][LDS_ITEMS_CAP];
    uint32_t bx = blockIdx.x;
    if (MERGED) {
        const uint32_t per_chunk = 8u * (uint32_t)P.merge_gather_n;
        const uint32_t n_pairs = (uint32_t)(P.n_panos * P.n_yaw);
        const uint32_t gchunks = (n_pairs + (uint32_t)P.gather_ppb - 1u) / (uint32_t)P.gather_ppb;
        const uint32_t gw = per_chunk * gchunks;
        if (bx < gw) {
            if (blockIdx.y != 0)
                return;
            const uint32_t gchunk = bx / per_chunk;
            int pitch_i, tile_id;
            if (!tile_of_list<true>(P, P.merge_gather_list, AUD_GATHER_LIST, pitch_i, tile_id, bx - gchunk * per_chunk, P.merge_gather_n))
                return;
            const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
            const PieceHdr h = P.hdr[(size_t)pitch_i * tiles + tile_id];
            const TileGeo G = tile_geo(P, h, pitch_i, tile_id, (int)threadIdx.x);
            if (G.mode != 2)
                return;
            static_assert(sizeof(tile4) >= (VIEWS_BLOCK / 64) * VIEWS_PXT * 64 * sizeof(uint32_t), "the gather body's staging dwords");
            draw_gather<true>(P, src, ydesc, out, G, reinterpret_cast<uint32_t*>(&tile4[0][0]), (int)gchunk);
            return;
        }
        bx -= gw;
    }
    const uint32_t xcd = bx & 7u, q = bx >> 3;
    const uint32_t n_tiles = (uint32_t)P.band_tiles;
    uint32_t first = info->first[xcd], last = info->first[xcd + 1];
    const bool rev = info->reversed[xcd] != 0u;
    last = last < n_tiles ? last : n_tiles;  // (whatever the table holds, a valid tile is drawn)
    first = first < last ? first : last;
    const uint32_t L = last - first;
    int chunk = (int)blockIdx.y, ppb = 0;
    uint32_t e = q;
    if (P.band_tail > 0) {
        const uint32_t K = (uint32_t)P.band_tail < L ? (uint32_t)P.band_tail : L;
        if (q >= L - K) {
            const uint32_t i = q - (L - K), parts = (uint32_t)P.main_tail_parts;
            if (i >= parts * K)
                return;
            e = L - K + i / parts;
            chunk = (int)(i - (i / parts) * parts);
            ppb = (P.n_panos * P.n_yaw + (int)parts - 1) / (int)parts;
        }
    }
    if (e >= L)
        return;
    const uint32_t tile = rev ? last - 1u - e : first + e;
    P2P_AUD_LT(P.audit, AUD_BAND_TILE, tile, n_tiles);
    const PieceHdr h = hdr[tile];
    TileGeo G{};
    G.mode = 1;
    G.n_items = (int)(h.mode_items >> 8);
    G.c0 = h.c0;
    G.c1 = h.c1;
    G.band_r0 = (int)(h.rows & 0xFFFFu);
    G.band_row_items = (int)(h.rows >> 16);
    G.band_row_items = G.band_row_items > 0 ? G.band_row_items : 1;
    G.slot = tile;
    P2P_AUD_LT(P.audit, AUD_MAIN_HDR, G.n_items, LDS_ITEMS_CAP + 1);
    P2P_AUD_LT(P.audit, AUD_BAND_RECT, (uint32_t)G.band_row_items * (uint32_t)((G.n_items + G.band_row_items - 1) / G.band_row_items), LDS_ITEMS_CAP + 1);
    draw_tight<true, MASKED>(P, src, ydesc, out, G, px + (size_t)tile * (VIEWS_BLOCK * VIEWS_PXT), nullptr, tile4, nullptr,
                     chunk * (TILE_W == 128 && P.main_span > 1 ? P.main_span : 1), ppb, grp + (size_t)tile * VIEWS_BLOCK);
}

#ifndef P2P_DIRECT_WAVES
#define P2P_DIRECT_WAVES 5
#endif
template <int BORDER>
__global__ __launch_bounds__(VIEWS_BLOCK, P2P_DIRECT_WAVES) void remap_views_table_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, const uint32_t* __restrict__ ytab, uint8_t* __restrict__ out,
    const PieceHdr* __restrict__ hdr, const uint32_t* __restrict__ gather_list)
{
    int pitch_i, tile_id;
    tile_of_list<false>(P, gather_list, AUD_TABLE_LIST, pitch_i, tile_id);
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const PieceHdr h = hdr[(size_t)pitch_i * tiles + tile_id];
    const TileGeo G = tile_geo(P, h, pitch_i, tile_id, (int)threadIdx.x);
    draw_rest<true, BORDER>(P, src, ytab, nullptr, nullptr, out, G, nullptr, nullptr, nullptr);
}

// which = 0: the main kernel, 1: the rest, 2: the table kernel, 3: the gather kernel (they write disjoint pixels; the
// host launches all but the first only when the plan or the yaw tables have something for them)
hipError_t launch_remap_views(const ViewsParams& P, int which, hipStream_t st)
{
    const int n_pairs = P.n_panos * P.n_yaw;
    if (which == 2 || which == 3) {
        const int np = (which == 2 && P.use_pair_list) ? P.n_odd_pairs : n_pairs;
        const dim3 grid(which == 3 ? 8 * P.n_list : P.n_gather, 1, (np + P.gather_ppb - 1) / P.gather_ppb);
#define P2P_LAUNCH_TABLE(B) \
    hipLaunchKernelGGL(remap_views_table_kernel<B>, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ytab, P.out, P.hdr, P.gather_list)
        if (which == 2) {
            switch (P.border) {
            case 1: P2P_LAUNCH_TABLE(1); break;
            case 2: P2P_LAUNCH_TABLE(2); break;
            case 3: P2P_LAUNCH_TABLE(3); break;
            case 4: P2P_LAUNCH_TABLE(4); break;
            default: P2P_LAUNCH_TABLE(0); break;
            }
        }
#undef P2P_LAUNCH_TABLE
        else
            hipLaunchKernelGGL(remap_views_gather_kernel, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ydesc, P.out, P.hdr, P.gather_list);
        return hipGetLastError();
    }
    if (which == 4) {  // the band kernel: per XCD band_per list entries (+ the split tail), x chunks of pairs
        int zb = (n_pairs + P.pairs_per_block - 1) / P.pairs_per_block;
        if (TILE_W == 128 && P.main_span > 1)
            zb = (zb + P.main_span - 1) / P.main_span;
        const bool merged = P.merge_gather_list != nullptr && P.merge_gather_n > 0;
        const int gw = merged ? 8 * P.merge_gather_n * ((n_pairs + P.gather_ppb - 1) / P.gather_ppb) : 0;
        const dim3 grid(gw + 8 * (P.band_per + (P.band_tail > 0 ? (P.main_tail_parts - 1) * P.band_tail : 0)), P.band_tail > 0 ? 1 : zb, 1);
#define P2P_LAUNCH_BAND(MASKED, MERGED)                                                                                         \
        hipLaunchKernelGGL((remap_views_band_kernel<MASKED, MERGED>), grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ydesc, P.out, \
                           P.band_hdr, P.band_px, P.band_grp, P.band_info)
        if (P.view_mask) {
            if (merged) P2P_LAUNCH_BAND(true, true); else P2P_LAUNCH_BAND(true, false);
        } else {
            if (merged) P2P_LAUNCH_BAND(false, true); else P2P_LAUNCH_BAND(false, false);
        }
#undef P2P_LAUNCH_BAND
        return hipGetLastError();
    }
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    int zblocks = (n_pairs + P.pairs_per_block - 1) / P.pairs_per_block;
    if (which == 1 && P.use_pair_list)
        zblocks = (P.n_odd_pairs + P.rest_ppb - 1) / P.rest_ppb;
    // 8 XCDs, each a contiguous run of tiles; (tile, chunk, pitch view): see pair_chunk
    if (which == 0 && TILE_W == 128 && P.main_span > 1)  // a main-kernel workgroup loops over main_span chunks
        zblocks = (zblocks + P.main_span - 1) / P.main_span;
    dim3 grid(8 * ((tiles + 7) / 8), P.chunk_outer ? P.n_pitch : zblocks, P.chunk_outer ? zblocks : P.n_pitch);
    if (which == 0 && TILE_W != 128 && P.main_list && P.main_tail > 0)  // list order, one chunk: every entry once, the last main_tail of every XCD twice
        grid = dim3(8 * (P.main_stride + (P.main_tail_parts - 1) * P.main_tail), 1, 1);
    else if (which == 0 && P.main_list)  // list order: (blocks of main_group entries) x chunks, per XCD
        grid = dim3(8 * ((P.main_stride + P.main_group - 1) / P.main_group) * (P.main_group * P.main_chunks + (P.pf_lead > 0 ? 1 : 0)), 1, 1);
    else if (which == 0 && P.pf_lead > 0)  // one table-prefetch workgroup in PF_GROUP + 1 (p2p_tile.h: main_block_role)
        grid.x = 8 * (((tiles + 7) / 8 + PF_GROUP - 1) / PF_GROUP) * (PF_GROUP + 1);
    if (which == 0 && P.main_list && P.merge_gather_list && P.merge_gather_n > 0) {
        grid.x += 8 * P.merge_gather_n * ((n_pairs + P.gather_ppb - 1) / P.gather_ppb);
        hipLaunchKernelGGL(remap_views_kernel<true>, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ydesc, P.out, P.hdr, P.px, P.items, P.main_count);
    } else if (which == 0)
        hipLaunchKernelGGL(remap_views_kernel<false>, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ydesc, P.out, P.hdr, P.px, P.items, P.main_count);
    else
        hipLaunchKernelGGL(remap_views_rest_kernel, grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ytab, P.ydesc, P.f4tab,
                           P.out, P.hdr, P.px, P.items);
    return hipGetLastError();
}

}  // namespace P2P_SHAPE_NS
}  // namespace p2p
