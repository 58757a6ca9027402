// p2p_views.hip -- the hot kernel: both cv2.remap stages of every (panorama, yaw, pitch) view in one launch
//   cv2.remap x2       P:192-199, P:212-218 -> remap_views_kernel (both stages fused, fixed point; modes: draw,
//                                            plan, sub-tile); 3-channel single remaps of the legacy tool (L:179) too
// Reference behaviour (cited, never copied):
//   P = /root/reference/app/panorama_to_plane-pitch.py, L = /root/reference/app/legacy/panorama_to_plane.py
// The fixed-point arithmetic is OpenCV 4.10's (imgwarp.cpp remapBilinear, INTER_BITS = 5,
// INTER_REMAP_COEF_BITS = 15); see DESIGN.md "Arithmetic contract".
// Compiled with -ffp-contract=off: every float operation below rounds where NumPy rounds.
#include "p2p_inline.h"

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

struct __attribute__((aligned(4))) Q16 { uint32_t d[4]; };

// ---- diagnostic build only: in-kernel phase stamps (never compiled into the shipped library) ----
__device__ unsigned long long g_stamps[8 * 4096];  // [counter][slot]: spread, same-address atomics crawl
#ifdef P2P_STAMPS
#define STAMP(var)                                                                              \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// per-pixel stage-2 weights, constant across (panorama, yaw) pairs
struct TapWeights {
    uint32_t gx2, fx2;  // [32-fx, 32-fx], [fx, fx] as two u16
    uint32_t wy;        // 64 * [32-fy, fy] as two u16 (0 for a pixel with no footprint)
};

// Stage 2 with packed 16-bit maths: per channel, the two rows ride in the two halves of a dword:
//   H = [a.c, c.c] * [gx, gx] + [b.c, d.c] * [fx, fx]   (v_pk_mul_lo_u16, v_pk_mad_u16; <= 8160)
//   V = H.lo * gy + H.hi * fy + 512                      (v_dot2_u32_u16)
// identical in value to blend4().
__device__ __forceinline__ uint32_t blend4_packed(uint32_t a, uint32_t b, uint32_t c, uint32_t d,
                                                  const TapWeights& w)
{
    const u16x2 gx = as_u16x2(w.gx2), fx = as_u16x2(w.fx2), wy = as_u16x2(w.wy);
    // v_perm_b32(S0, S1, sel): selector bytes 0-3 pick S1's bytes, 4-7 pick S0's, 0x0c is zero
    const u16x2 b_ac = as_u16x2(__builtin_amdgcn_perm(c, a, 0x0C040C00u));
    const u16x2 b_bd = as_u16x2(__builtin_amdgcn_perm(d, b, 0x0C040C00u));
    const u16x2 g_ac = as_u16x2(__builtin_amdgcn_perm(c, a, 0x0C050C01u));
    const u16x2 g_bd = as_u16x2(__builtin_amdgcn_perm(d, b, 0x0C050C01u));
    const u16x2 r_ac = as_u16x2(__builtin_amdgcn_perm(c, a, 0x0C060C02u));
    const u16x2 r_bd = as_u16x2(__builtin_amdgcn_perm(d, b, 0x0C060C02u));
    const u16x2 hb = b_ac * gx + b_bd * fx;
    const u16x2 hg = g_ac * gx + g_bd * fx;
    const u16x2 hr = r_ac * gx + r_bd * fx;
    // wy holds 64*[32-fy, fy]: the sums come out scaled by 64, so (sum + 512) >> 10 is byte 2 of each
    const uint32_t vb = __builtin_amdgcn_udot2(hb, wy, 32768u, false);
    const uint32_t vg = __builtin_amdgcn_udot2(hg, wy, 32768u, false);
    const uint32_t vr = __builtin_amdgcn_udot2(hr, wy, 32768u, false);
    const uint32_t bg = __builtin_amdgcn_perm(vg, vb, 0x0C0C0602u);  // B | G << 8
    return __builtin_amdgcn_perm(vr, bg, 0x0C060100u);               // | R << 16
}

// ---------------------------------------------------------------------------------------------
// The hot kernel.  One workgroup = one TILE_W x TILE_H tile of output pixels of one pitch view,
// VIEWS_PXT pixels per thread.  Once per tile: every thread evaluates (or loads) the pitch-stage
// coordinates of its pixels and quantises them as cv::remap does; the workgroup reduces the
// tile's footprint [c0..c1+1] x [r0..r1+1] in the yaw-resampled panorama ("rot") and cuts it into
// items of 4 horizontally adjacent rot pixels.  Then, per (panorama, yaw) pair of its chunk:
//   stage 1  the yaw map is a circular column shift (YawDesc), so a footprint row is one
//            contiguous run of source bytes: each thread loads one 4-byte-aligned 16-byte piece
//            (5 1/3 source pixels: fully coalesced, no per-pixel table lookup), blends 4 rot
//            pixels in registers with the exact uint8 arithmetic and writes them to the LDS tile
//            (double-buffered) with one ds_write_b128; the loads of the NEXT pair are issued
//            before stage 2 so that their latency hides behind it;
//   stage 2  after one barrier each thread reads the 2x2 taps of its pixels from LDS, blends
//            with cv::remap's fixed-point weights, and the tile is stored as aligned dwords.
// Whatever does not fit that scheme takes the direct path (same arithmetic, taps gathered from
// global memory through the packed yaw table): footprints too large for LDS or touching the
// panorama border (views containing a pole), panorama widths not divisible by 4, yaw rows that
// are not a shift.  Blocks map to tiles XCD-aware: each of the 8 XCDs owns a contiguous run of
// the tile raster, so neighbouring tiles (shared source halo and output lines) meet in one L2.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(as_u16x2(a), as_u16x2(b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(as_u16x2(a), as_u16x2(b)));
}

// Wave-wide packed-u16 min / max; the result is valid in lane 63.  gfx9 DPP: two quad permutes, the
// two row mirrors, then row_bcast15 / row_bcast31 carry the row results up to the last row.
template <bool MIN>
__device__ __forceinline__ uint32_t wave_reduce_pk(uint32_t v)
{
    const int ident = MIN ? -1 : 0;
#define P2P_STEP(ctrl, rmask)                                                                        \
    {                                                                                               \
        uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, ctrl, rmask, 0xF, false); \
        v = MIN ? pk_min(v, o) : pk_max(v, o);                                                      \
    }
    P2P_STEP(0xB1, 0xF)   // quad_perm [1,0,3,2]
    P2P_STEP(0x4E, 0xF)   // quad_perm [2,3,0,1]
    P2P_STEP(0x141, 0xF)  // row_half_mirror
    P2P_STEP(0x140, 0xF)  // row_mirror
    P2P_STEP(0x142, 0xA)  // row_bcast15 -> rows 1 and 3
    P2P_STEP(0x143, 0xC)  // row_bcast31 -> rows 2 and 3
#undef P2P_STEP
    return v;
}

struct PairCtx {      // uniform per (tile, pair); precomputed per lane at tile set-up, read back with v_readlane
    bool fast;        // LDS scheme applies (the yaw row is a circular shift)
    bool per_column;  // per-column weights (f4tab) instead of one f
    int joff;         // tile column of rot column c0
    uint32_t goff;    // byte offset of the first item of a footprint row within a source row
    uint32_t wrap_g;  // items with g >= wrap_g wrap to the start of the row
    uint32_t f;       // uniform weight
    int cf0;          // rot column of source column 4 * g0
    int yaw_i;
    int pano;         // panorama index
    int korig;        // pair index inside the chunk (its output slot is pair0 + korig)
};

// MAPSRC: 0 = coordinates computed in-kernel (pitch_map_eval), 1 = caller float maps, 2 = the job's
// coordinate cache (what an earlier MAPSRC 0 launch stored; the reference's pitch_mapping_cache, P:62-73)
// MODE: 0 = the view kernel proper (32x16 tiles, 2 pixels per thread); 1 = plan: classify the tiles whose
// footprint outgrows the LDS buffers and list their sub-tiles (once per job geometry, nothing is drawn);
// 2 = sub-tile pass: one listed 32x8 or 16x8 sub-tile per workgroup, 1 pixel per thread
template <int MAPSRC, int MODE>
__device__ __forceinline__ void views_body(
    const ViewsParams& P, const uint8_t* __restrict__ src, const uint32_t* __restrict__ ytab,
    const YawDesc* __restrict__ ydesc, const uint32_t* __restrict__ f4tab,
    const PitchConst* __restrict__ pitch, const float* __restrict__ mapU,
    const float* __restrict__ mapV, uint8_t* __restrict__ out, int32_t* __restrict__ coords,
    uint4 (*tile4)[LDS_ITEMS_CAP], int* bbox, uint32_t (*half_box)[4], const int bx, const int gx)
{
    constexpr int PXT = MODE == 2 ? 1 : VIEWS_PXT;
    const int t = threadIdx.x;
    int pitch_i, x0, y0, sub_w = TILE_W, plan_slot = 0;
    if (MODE == 2) {
        // the sub-tile workgroups of view row y: entries y * gx .. y * gx + gx - 1 of the plan
        const int ei = (int)blockIdx.y * gx + bx;
        if (ei >= P.plan_n)
            return;
        const uint2 e = P.plan[ei];  // x0 | y0 << 15 | (16-wide) << 30, pitch index
        x0 = (int)(e.x & 0x7FFFu);
        y0 = (int)((e.x >> 15) & 0x7FFFu);
        sub_w = (e.x >> 30) ? TILE_W / 2 : TILE_W;
        pitch_i = (int)e.y;
    } else {
        const int tiles_x = (P.ow + TILE_W - 1) / TILE_W;
        const int tiles_y = (P.oh + TILE_H - 1) / TILE_H;
        const int chunk = gx >> 3;  // gx == 8 * ceil(tiles / 8)
        const int tile_id = (bx & 7) * chunk + (bx >> 3);
        if (tile_id >= tiles_x * tiles_y)
            return;
        // heaviest views first (the host orders pitch_order by |pitch - 90| descending): a smoother tail
        pitch_i = P.pitch_order[blockIdx.y];
        x0 = (tile_id % tiles_x) * TILE_W;
        y0 = (tile_id / tiles_x) * TILE_H;
        plan_slot = pitch_i * tiles_x * tiles_y + tile_id;
        // a tile the plan lists is drawn by the sub-tile workgroups: each pixel's map is evaluated by exactly
        // one compiled instance of pitch_map_eval (two inlined copies can differ in the last bit)
        if (MODE == 0 && P.use_plan && P.plan_flag[plan_slot])
            return;
    }
    const int px = x0 + (MODE == 2 ? t % sub_w : t % TILE_W);
    const int py0 = y0 + (MODE == 2 ? t / sub_w : t / TILE_W);
    constexpr int ROWSTEP = VIEWS_BLOCK / TILE_W;  // rows between a thread's pixels
    constexpr int SUB_H = TILE_H / 2;              // sub-tiles are 32x8 or 16x8

    // ---- pitch-stage coordinates of this thread's pixels, quantised as cv::remap does ----
    int ix[PXT], iy[PXT];
    uint32_t fx[PXT], fy[PXT];
    bool inside[PXT], live[PXT], inrange[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const int py = py0 + j * ROWSTEP;
        inside[j] = px < P.ow && py < P.oh && (MODE != 2 || py < y0 + SUB_H);
        int sx = INT32_MIN, sy = INT32_MIN;
        if (inside[j]) {
            const size_t k = ((size_t)pitch_i * P.oh + py) * P.ow + px;
            if (MAPSRC == 2) {
                // streamed once per launch: non-temporal, so that the 8 bytes per pixel do not evict source lines
                const long long sc = __builtin_nontemporal_load(reinterpret_cast<const long long*>(coords) + k);
                sx = (int)(uint32_t)sc;
                sy = (int)(sc >> 32);
            } else {
                float U, V;
                if (MAPSRC == 1) {
                    U = mapU[k];
                    V = mapV[k];
                } else {
                    PitchConst pc = pitch[pitch_i];
                    pitch_map_eval((float)px, (float)py, P.geom, pc.c, pc.s, U, V);
                }
                sx = cv_round_f32(U * 32.0f);
                sy = cv_round_f32(V * 32.0f);
                if (coords && blockIdx.z == 0)
                    reinterpret_cast<int2*>(coords)[k] = make_int2(sx, sy);
            }
        }
        ix[j] = sat_short(sx >> 5);
        iy[j] = sat_short(sy >> 5);
        fx[j] = (uint32_t)sx & 31u;
        fy[j] = (uint32_t)sy & 31u;
        // A pixel contributes only if its 2x2 footprint touches the panorama (BORDER_CONSTANT 0:
        // cv::remap writes borderValue when sx >= w || sx+1 < 0 || sy >= h || sy+1 < 0).  For the
        // reference's clipped maps that is every pixel except NaN ones (ix = iy = -32768).
        inrange[j] = inside[j] && ix[j] >= -1 && iy[j] >= -1 && ix[j] < P.pw && iy[j] < P.ph;
        // other border modes (legacy entry point, L:179) resolve every tap to some pixel
        live[j] = P.border == 0 ? inrange[j] : inside[j];
    }

    // ---- footprint of the tile in rot space: packed (ix+1, iy+1) u16 pairs, one min and one max
    // reduction per wave with DPP (v_pk_min_u16 / v_pk_max_u16), then across the 4 waves through LDS ----
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < PXT; ++j)
        if (inrange[j]) {
            const uint32_t key = (uint32_t)(ix[j] + 1) | (uint32_t)(iy[j] + 1) << 16;  // both in 0..32767
            kmin = pk_min(kmin, key);
            kmax = pk_max(kmax, key);
        }
    kmin = wave_reduce_pk<true>(kmin);
    kmax = wave_reduce_pk<false>(kmax);
    if ((t & 63) == 63) {
        bbox[(t >> 6) * 2] = (int)kmin;
        bbox[(t >> 6) * 2 + 1] = (int)kmax;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < VIEWS_BLOCK / 64; ++w) {
        kmin = w == 0 ? (uint32_t)bbox[0] : pk_min(kmin, (uint32_t)bbox[2 * w]);
        kmax = w == 0 ? (uint32_t)bbox[1] : pk_max(kmax, (uint32_t)bbox[2 * w + 1]);
    }
    const int c0 = (int)(kmin & 0xFFFFu) - 1, r0 = (int)(kmin >> 16) - 1;
    const int c1 = kmax ? (int)(kmax & 0xFFFFu) - 1 : -2, r1 = kmax ? (int)(kmax >> 16) - 1 : -2;
    const bool any_live = c1 >= -1;
    // footprint -> 4-pixel items per row (+3: alignment slack; +1 column and row for the right / lower taps)
    auto items_of = [](int a0, int a1, int b0, int b1, int& g) {
        g = (a1 - a0 + 2 + 6) >> 2;
        return (b1 - b0 + 2) * g;
    };
    int G;
    const int items = items_of(c0, c1, r0, r1, G);
    const int rowdw = 4 * G;  // LDS tile row stride in dwords
    // the LDS scheme needs the whole footprint strictly inside the panorama (so that no tap is a
    // border tap) and a width divisible by 4 (so that 12-byte items never straddle a row end)
    // with a non-constant border a pixel outside the panorama still reads pixels (reflected, wrapped ...):
    // such tiles go the direct way, where the taps are resolved by cv::borderInterpolate
    bool stray = false;
    if (P.border != 0) {
        bool mine = false;
#pragma unroll
        for (int j = 0; j < PXT; ++j)
            mine |= inside[j] && !inrange[j];
        stray = __syncthreads_or(mine) != 0;
    }
    const bool lds_ok = any_live && !stray && (P.pw & 3) == 0 && c0 >= 0 && r0 >= 0 && c1 + 1 < P.pw &&
                        r1 + 1 < P.ph;
    const bool fast_tile = lds_ok && G <= 255 && items <= LDS_ITEMS_CAP;

    // A tile that is fine except that its footprint outgrows the LDS buffers (views towards a pole: the rows
    // stretch by 1 / sin(theta)) is drawn by the sub-tile pass instead: as two 32x8 halves if both fit, else
    // as four 16x8 quarters (each of which decides again between the LDS scheme and direct gathers).
    if (MODE == 1) {
        if (lds_ok && !fast_tile) {
            // half_box: per half (pixel j): min / max of ix + 1, min / max of iy + 1
            if (t < 2) {
                half_box[t][0] = 0xFFFFu; half_box[t][1] = 0u; half_box[t][2] = 0xFFFFu; half_box[t][3] = 0u;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                if (inrange[j]) {
                    atomicMin(&half_box[j][0], (uint32_t)(ix[j] + 1));
                    atomicMax(&half_box[j][1], (uint32_t)(ix[j] + 1));
                    atomicMin(&half_box[j][2], (uint32_t)(iy[j] + 1));
                    atomicMax(&half_box[j][3], (uint32_t)(iy[j] + 1));
                }
            __syncthreads();
            // a half is fine if its footprint rectangle fits, or if the items its taps can touch do (the
            // sub-tile pass then keeps a compacted item list, see there); count those with the same bitmap
            bool halves = true;
            for (int h = 0; h < 2; ++h) {
                if (half_box[h][1] == 0u)
                    continue;  // no live pixel
                const int hc0 = (int)half_box[h][0] - 1, hr0 = (int)half_box[h][2] - 1;
                int g;
                const int n = items_of(hc0, (int)half_box[h][1] - 1, hr0, (int)half_box[h][3] - 1, g);
                if (g <= 255 && n <= LDS_ITEMS_CAP)
                    continue;
                if (n > 65536 || g >= 65536) {
                    halves = false;
                    continue;
                }
                uint32_t* bm = reinterpret_cast<uint32_t*>(&tile4[0][0]);
                __syncthreads();
                for (int i = 0; i < 8; ++i)
                    bm[t * 8 + i] = 0u;
                if (t == 0)
                    half_box[h][1] = 0u;  // reused as the counter below (its value is already in n, g)
                __syncthreads();
#pragma unroll
                for (int j = 0; j < PXT; ++j)
                    if (j == h && inrange[j]) {
                        const uint32_t b0 = (uint32_t)((iy[j] - hr0) * g + ((ix[j] - hc0) >> 2));
                        for (int dr = 0; dr < 2; ++dr)
                            for (int dg = 0; dg < 2; ++dg) {
                                const uint32_t b = b0 + (uint32_t)(dr * g + dg);
                                atomicOr(&bm[b >> 5], 1u << (b & 31u));
                            }
                    }
                __syncthreads();
                uint32_t cnt = 0u;
                for (int i = 0; i < 8; ++i)
                    cnt += (uint32_t)__popc(bm[t * 8 + i]);
                atomicAdd(&half_box[h][1], cnt);
                __syncthreads();
                halves = halves && half_box[h][1] <= (uint32_t)LDS_ITEMS_CAP;
            }
            if (t == 0) {
                uint2 e[4];
                int n = 0;
                for (int sy = 0; sy < 2; ++sy)
                    for (int sx = 0; sx < (halves ? 1 : 2); ++sx) {
                        const int ex = x0 + sx * (TILE_W / 2), ey = y0 + sy * SUB_H;
                        if (ex < P.ow && ey < P.oh)
                            e[n++] = make_uint2((uint32_t)ex | (uint32_t)ey << 15 | (halves ? 0u : 1u << 30), (uint32_t)pitch_i);
                    }
                const uint32_t base = atomicAdd(P.plan_count, (uint32_t)n);
                for (int i = 0; i < n; ++i)
                    P.plan[base + i] = e[i];
                P.plan_flag[plan_slot] = 1;
            }
        }
        return;
    }
    // output addressing: 4 horizontally adjacent pixels = 12 bytes = 3 aligned dwords
    const int lane4 = t & 3;
    const bool fast_store = (P.ow & 3) == 0;
    const size_t view_bytes = (size_t)P.oh * P.ow * 3;
    const uint32_t pix_off = (uint32_t)(((size_t)py0 * P.ow + px) * 3);  // < 3 * 32766^2 < 2^32
    const uint32_t pix_step = (uint32_t)ROWSTEP * (uint32_t)P.ow * 3u;
    // dword lane4 of the 12 bytes P0 P1 P2 P3: bytes of the own pixel (0-2) and of the next lane's (4-6)
    const uint32_t store_sel = lane4 == 0 ? 0x04020100u : (lane4 == 1 ? 0x05040201u : 0x06050402u);

    const int pair0 = blockIdx.z * P.pairs_per_block;
    int pair1 = pair0 + P.pairs_per_block;
    const int n_pairs = P.n_panos * P.n_yaw;
    if (pair1 > n_pairs)
        pair1 = n_pairs;
    // pair -> panorama: multiply-high by ceil(2^32 / n_yaw) (exact for the job's sizes, host check); with one yaw
    // the constant would be 2^32, which does not fit, and the pair index is the panorama index anyway
    auto pano_of = [&](int pair) {
        return P.n_yaw == 1 ? pair : (int)__umulhi((uint32_t)pair, P.n_yaw_magic);
    };
    int pano_i = pano_of(pair0);
    int yaw_i = pair0 - pano_i * P.n_yaw;

    auto store_pixels = [&](int pair, const uint32_t (&pix)[PXT]) {
        // [pano][yaw][pitch][oh][ow][3]
        uint8_t* O = out + ((size_t)pair * P.n_pitch + pitch_i) * view_bytes;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const uint32_t off = pix_off + (uint32_t)j * pix_step;
            if (fast_store) {
                // lanes 4k..4k+3 hold pixels P0..P3; lanes with lane4 < 3 emit dword lane4 of the 12 bytes
                // neighbour lane's pixel: row_shl:1 DPP (lane4 groups never straddle a 16-lane row)
                uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pix[j], 0x101, 0xF, 0xF, true);
                uint32_t dw = __builtin_amdgcn_perm(nxt, pix[j], store_sel);
                // the byte offset stays a 32-bit VGPR next to the scalar view base (saddr store form): the
                // empty asm keeps the compiler from hoisting a 64-bit copy of it out of the pair loop
                uint32_t voff = off + (uint32_t)lane4;
                asm volatile("" : "+v"(voff));
                if (inside[j] && lane4 < 3)
                    *reinterpret_cast<uint32_t*>(O + voff) = dw;
            } else if (inside[j]) {
                uint8_t* o = O + off;
                o[0] = (uint8_t)pix[j];
                o[1] = (uint8_t)(pix[j] >> 8);
                o[2] = (uint8_t)(pix[j] >> 16);
            }
        }
    };

    auto direct_pixels = [&](const uint8_t* __restrict__ S, int yi, uint32_t (&pix)[PXT]) {
        // same arithmetic, taps gathered from global memory through the packed yaw table
        const uint32_t* __restrict__ T = ytab + (size_t)yi * P.pw;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            pix[j] = 0;
            if (live[j] && P.border != 0) {
                const int xa = border_interpolate(ix[j], P.pw, P.border), xb = border_interpolate(ix[j] + 1, P.pw, P.border);
                const int ya = border_interpolate(iy[j], P.ph, P.border), yb = border_interpolate(iy[j] + 1, P.ph, P.border);
                const uint8_t* row0 = S + (size_t)ya * P.src_pitch;
                const uint8_t* row1 = S + (size_t)yb * P.src_pitch;
                const uint32_t t0 = T[xa], t1 = T[xb];
                pix[j] = blend4(rot_pixel(row0, t0), rot_pixel(row0, t1), rot_pixel(row1, t0), rot_pixel(row1, t1),
                                fx[j], fy[j]);
            } else if (live[j]) {
                const bool c0in = ix[j] >= 0, c1in = ix[j] + 1 < P.pw, r0in = iy[j] >= 0, r1in = iy[j] + 1 < P.ph;
                const uint8_t* row0 = S + (ptrdiff_t)iy[j] * P.src_pitch;
                const uint8_t* row1 = row0 + P.src_pitch;
                const uint32_t t0 = c0in ? T[ix[j]] : 0u, t1 = c1in ? T[ix[j] + 1] : 0u;
                uint32_t a = (c0in && r0in) ? rot_pixel(row0, t0) : 0u;
                uint32_t b = (c1in && r0in) ? rot_pixel(row0, t1) : 0u;
                uint32_t c = (c0in && r1in) ? rot_pixel(row1, t0) : 0u;
                uint32_t d = (c1in && r1in) ? rot_pixel(row1, t1) : 0u;
                pix[j] = blend4(a, b, c, d, fx[j], fy[j]);
            }
        }
    };

    // ---- sub-tile pass only: a footprint RECTANGLE too large for the LDS buffers is usually sparse (towards a
    // pole the rows stretch, neighbouring pixels hit items far apart).  Mark the items the taps can touch in a
    // bitmap (each pixel: items g0, g0 + 1 of rows rr, rr + 1 -- every alignment joff = 0..3 stays inside them),
    // rank them, and keep only those in LDS: slot k of the buffer holds the k-th needed item, and a pixel's two
    // consecutive items are consecutive slots, so the tap reads stay "two dwords at base + 4 * joff".
    // The tile buffers themselves serve as scratch: bitmap 8 KB | per-word ranks 4 KB | item list 2 KB | scan 1 KB.
    bool list_tile = false;
    int n_need = 0;
    uint32_t list_item[VIEWS_SLOTS] = {0u, 0u};
    uint32_t list_tap_up[PXT], list_tap_lo[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j)
        list_tap_up[j] = list_tap_lo[j] = 0u;
    if (MODE == 2) {
        constexpr int MAX_BITS = 65536, WPT = MAX_BITS / 32 / VIEWS_BLOCK;  // 8 bitmap words per thread
        const int Ht = r1 - r0 + 2;
        // (only when the rectangle does not fit: for rectangles that fit, building the list costs more than the
        // empty items it saves at 12 yaws per workgroup -- measured)
        if (lds_ok && !fast_tile && Ht * G <= MAX_BITS && G < 65536) {
            uint32_t* bm = reinterpret_cast<uint32_t*>(&tile4[0][0]);
            unsigned short* wpre = reinterpret_cast<unsigned short*>(bm + MAX_BITS / 32);
            uint32_t* lst = bm + MAX_BITS / 32 + MAX_BITS / 64;
            uint32_t* scan = lst + LDS_ITEMS_CAP;
            const int nw = (Ht * G + 31) >> 5;
#pragma unroll
            for (int i = 0; i < WPT; ++i)
                bm[t * WPT + i] = 0u;
            __syncthreads();
            uint32_t bit0[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                bit0[j] = 0u;
                if (live[j]) {
                    bit0[j] = (uint32_t)((iy[j] - r0) * G + ((ix[j] - c0) >> 2));
                    for (int dr = 0; dr < 2; ++dr)
                        for (int dg = 0; dg < 2; ++dg) {
                            const uint32_t b = bit0[j] + (uint32_t)(dr * G + dg);
                            atomicOr(&bm[b >> 5], 1u << (b & 31u));
                        }
                }
            }
            __syncthreads();
            // ranks: per-thread popcount of its 8 words, block-wide exclusive scan, then per word
            uint32_t wv[WPT];
            uint32_t mine = 0u;
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                wv[i] = t * WPT + i < nw ? bm[t * WPT + i] : 0u;
                mine += (uint32_t)__popc(wv[i]);
            }
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if ((t & 63) >= d)
                    incl += up;
            }
            if ((t & 63) == 63)
                scan[t >> 6] = incl;
            __syncthreads();
            uint32_t base = incl - mine;
            for (int w = 0; w < (t >> 6); ++w)
                base += scan[w];
            n_need = (int)(scan[0] + scan[1] + scan[2] + scan[3]);
            if (n_need <= LDS_ITEMS_CAP) {
#pragma unroll
                for (int i = 0; i < WPT; ++i) {
                    if (t * WPT + i < nw)
                        wpre[t * WPT + i] = (unsigned short)base;
                    uint32_t v = wv[i];
                    while (v) {
                        const int b = __ffs((int)v) - 1;
                        v &= v - 1u;
                        lst[base++] = (uint32_t)((t * WPT + i) * 32 + b);
                    }
                }
                __syncthreads();
                auto rank_of = [&](uint32_t b) {
                    return (uint32_t)wpre[b >> 5] + (uint32_t)__popc(bm[b >> 5] & ((1u << (b & 31u)) - 1u));
                };
#pragma unroll
                for (int j = 0; j < PXT; ++j)
                    if (live[j]) {
                        const uint32_t within = 4u * (uint32_t)((ix[j] - c0) & 3);
                        list_tap_up[j] = 16u * rank_of(bit0[j]) + within;
                        list_tap_lo[j] = 16u * rank_of(bit0[j] + (uint32_t)G) + within;
                    }
#pragma unroll
                for (int k = 0; k < VIEWS_SLOTS; ++k)
                    list_item[k] = t + k * VIEWS_BLOCK < n_need ? lst[t + k * VIEWS_BLOCK] : lst[0];
                list_tile = true;
            }
            __syncthreads();  // the scratch becomes tile storage again
        }
    }

    if (!fast_tile && !list_tile) {
        for (int pair = pair0; pair < pair1; ++pair) {
            uint32_t pix[PXT];
            direct_pixels(src + (size_t)pano_i * P.pano_stride, yaw_i, pix);
            store_pixels(pair, pix);
            if (++yaw_i == P.n_yaw) {
                yaw_i = 0;
                ++pano_i;
            }
        }
        return;
    }

    // ---- LDS scheme: the items this thread produces (same for every pair) ----
    // A wave runs slot k only if its first lane has an item there (wave-uniform test).
    const int wave_base = __builtin_amdgcn_readfirstlane(t & ~63);
    uint32_t slot_off[VIEWS_SLOTS];  // (r0 + rr) * src_pitch + 12 * g
    uint32_t slot_g[VIEWS_SLOTS];
    const int n_items = (MODE == 2 && list_tile) ? n_need : items;  // LDS slots in use
    if (MODE == 2 && list_tile) {
#pragma unroll
        for (int k = 0; k < VIEWS_SLOTS; ++k) {  // the k-th needed item: rectangle index rr * G + g
            const uint32_t rr = list_item[k] / (uint32_t)G;
            slot_g[k] = list_item[k] - rr * (uint32_t)G;
            slot_off[k] = (uint32_t)(r0 + (int)rr) * (uint32_t)P.src_pitch + 12u * slot_g[k];
        }
    } else {
        // item / G by multiply-shift: exact for item * G < 2^20 (item < 512, G < 256)
        const uint32_t magic = ((1u << 20) + (uint32_t)G - 1u) / (uint32_t)G;
#pragma unroll
        for (int k = 0; k < VIEWS_SLOTS; ++k) {
            int item = t + k * VIEWS_BLOCK;
            if (item >= items)
                item = 0;  // surplus lanes redo item 0 into LDS space nobody reads
            const uint32_t rr = ((uint32_t)item * magic) >> 20;
            slot_g[k] = (uint32_t)item - rr * (uint32_t)G;
            slot_off[k] = (uint32_t)(r0 + (int)rr) * (uint32_t)P.src_pitch + 12u * slot_g[k];
        }
    }
    int tap[PXT];
    TapWeights tw[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        tap[j] = live[j] ? (iy[j] - r0) * rowdw + (ix[j] - c0) : 0;
        const uint32_t gx = 32u - fx[j], gy = 32u - fy[j];
        tw[j].gx2 = gx | (gx << 16);
        tw[j].fx2 = fx[j] | (fx[j] << 16);
        // a pixel with no footprint in the panorama (NaN coordinate) gets weight 0 everywhere:
        // (0 + 512) >> 10 == 0, the BORDER_CONSTANT value
        tw[j].wy = live[j] ? 64u * (gy | (fy[j] << 16)) : 0u;
    }
    const uint32_t row_bytes = 3u * (uint32_t)P.pw;
    const int ngroups = P.pw >> 2;

    // ---- per-pair contexts: lane k of every wave works out pair0 + k once; the loop reads them
    // back with v_readlane, so no descriptor load sits on the per-pair critical path ----
    uint32_t cw0 = 0, cw1 = 0;
    int cw2 = 0, cw3 = 0;
    bool ctx_plain = true;  // this lane's pair: circular-shift yaw with one weight for the whole tile
    {
        const int k = t & 63;
        if (k < pair1 - pair0) {
            // pair -> (panorama, yaw) by the host's multiply-high constant (exact for the job's sizes)
            cw3 = pano_of(pair0 + k);
            const int yi = pair0 + k - cw3 * P.n_yaw;
            const YawDesc yd = ydesc[yi];
            int i_first = c0 + yd.s;
            if (i_first >= P.pw)
                i_first -= P.pw;
            const int g0 = i_first >> 2;
            // uniform weight unless this yaw flickers or the tile holds the column clipped to pw-1
            const bool per_column = yd.mode == 1 || (yd.c_clamp >= c0 && yd.c_clamp <= c1 + 1);
            cw0 = 12u * (uint32_t)g0 | (uint32_t)(i_first & 3) << 20 | (uint32_t)(yd.mode != 2) << 22 |
                  (uint32_t)per_column << 23 | (uint32_t)yd.f << 24;
            cw1 = (uint32_t)(ngroups - g0) | (uint32_t)yi << 16;
            cw2 = 4 * g0 - yd.s;
            cw3 |= k << 26;  // n_panos < 2^26 (host check): the chunk-local pair index rides along
            ctx_plain = yd.mode != 2 && !per_column;
        }
    }
    // Plain pairs (the common case) run in the tight loop below, the others in the general loop after it:
    // the contexts are sorted plain-first across the lanes, so that one odd yaw in a chunk (6 of the 360
    // one-degree yaws on 8192 columns have per-column weights) does not slow its whole chunk down.
    const int npairs = pair1 - pair0;
    int nplain;
    {
        const int k = t & 63;
        const bool valid = k < npairs;
        const unsigned long long plain_mask = __ballot(valid && ctx_plain);
        const unsigned long long other_mask = __ballot(valid && !ctx_plain);
        nplain = __popcll(plain_mask);
        if (other_mask != 0ull) {
            const unsigned long long below = (1ull << k) - 1ull;
            const int r = !valid ? k : (ctx_plain ? __popcll(plain_mask & below) : nplain + __popcll(other_mask & below));
            cw0 = (uint32_t)__builtin_amdgcn_ds_permute(4 * r, (int)cw0);
            cw1 = (uint32_t)__builtin_amdgcn_ds_permute(4 * r, (int)cw1);
            cw2 = __builtin_amdgcn_ds_permute(4 * r, cw2);
            cw3 = __builtin_amdgcn_ds_permute(4 * r, cw3);
        }
    }
    auto pair_ctx = [&](int k) {
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)cw0, k);
        const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)cw1, k);
        PairCtx c;
        c.goff = w0 & 0xFFFFFu;
        c.joff = (int)((w0 >> 20) & 3u);
        c.fast = (w0 >> 22) & 1u;
        c.per_column = (w0 >> 23) & 1u;
        c.f = w0 >> 24;
        c.wrap_g = w1 & 0xFFFFu;
        c.yaw_i = (int)(w1 >> 16);
        c.cf0 = __builtin_amdgcn_readlane(cw2, k);
        const int w3 = __builtin_amdgcn_readlane(cw3, k);
        c.pano = w3 & 0x3FFFFFF;
        c.korig = (int)((uint32_t)w3 >> 26);
        return c;
    };

    auto issue_loads = [&](const PairCtx& pc, const uint8_t* __restrict__ S, Q16 (&q)[VIEWS_SLOTS],
                           uint32_t (&fw)[VIEWS_SLOTS]) {
#pragma unroll
        for (int k = 0; k < VIEWS_SLOTS; ++k) {
            if (wave_base + k * VIEWS_BLOCK < n_items) {
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

    if (nplain > 0) {
        // ---- tight loop: no per-pair mode branches, source pieces ping-pong between two register
        // sets (pair loop unrolled by two), so nothing is copied and nothing is re-decided per pair ----
        auto load_pieces = [&](int k, Q16 (&qq)[VIEWS_SLOTS]) {
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)cw0, k);
            const uint32_t wrap_g = (uint32_t)__builtin_amdgcn_readlane((int)cw1, k) & 0xFFFFu;
            const uint8_t* __restrict__ S = src + (size_t)(__builtin_amdgcn_readlane(cw3, k) & 0x3FFFFFF) * P.pano_stride;
            const uint32_t goff = w0 & 0xFFFFFu;
#pragma unroll
            for (int sl = 0; sl < VIEWS_SLOTS; ++sl)
                if (wave_base + sl * VIEWS_BLOCK < n_items) {
                    uint32_t off = slot_off[sl] + goff;
                    if (slot_g[sl] >= wrap_g)
                        off -= row_bytes;
                    qq[sl] = *reinterpret_cast<const Q16*>(S + off);
                }
        };
        auto stage1 = [&](int k, const Q16 (&qq)[VIEWS_SLOTS], uint4* tl4) {
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)cw0, k) >> 24;
#pragma unroll
            for (int sl = 0; sl < VIEWS_SLOTS; ++sl)
                if (wave_base + sl * VIEWS_BLOCK < n_items) {
                    // the piece holds source pixels 0..4 at byte offsets 0, 3, 6, 9, 12; one v_perm_b32
                    // both fetches a pixel across the dword seam and masks it
                    // (selector bytes 0-3 pick the second operand's bytes, 4-7 the first's, 0x0c is zero)
                    const uint32_t d0 = qq[sl].d[0], d1 = qq[sl].d[1], d2 = qq[sl].d[2], d3 = qq[sl].d[3];
                    uint4 o;
                    if (f != 0) {
                        const uint32_t f8 = 8u * f, g8 = 256u - f8;
                        const uint32_t m0 = d0 & 0x00FF00FFu, n0 = d0 & 0x0000FF00u;                    // bytes 0,1,2
                        const uint32_t m1 = __builtin_amdgcn_perm(d1, d0, 0x0C050C03u);                  // 3,(4),5
                        const uint32_t n1 = __builtin_amdgcn_perm(d1, d0, 0x0C0C040Cu);
                        const uint32_t m2 = __builtin_amdgcn_perm(d2, d1, 0x0C040C02u);                  // 6,(7),8
                        const uint32_t n2 = __builtin_amdgcn_perm(d2, d1, 0x0C0C030Cu);
                        const uint32_t m3 = __builtin_amdgcn_perm(d3, d2, 0x0C030C01u);                  // 9,(10),11
                        const uint32_t n3 = __builtin_amdgcn_perm(d3, d2, 0x0C0C020Cu);
                        const uint32_t m4 = d3 & 0x00FF00FFu, n4 = d3 & 0x0000FF00u;                    // 12,13,14
                        o.x = rot_blend8(m0, n0, m1, n1, f8, g8);
                        o.y = rot_blend8(m1, n1, m2, n2, f8, g8);
                        o.z = rot_blend8(m2, n2, m3, n3, f8, g8);
                        o.w = rot_blend8(m3, n3, m4, n4, f8, g8);
                    } else {
                        o.x = d0 & 0x00FFFFFFu;
                        o.y = __builtin_amdgcn_perm(d1, d0, 0x0C050403u);
                        o.z = __builtin_amdgcn_perm(d2, d1, 0x0C040302u);
                        o.w = __builtin_amdgcn_perm(d3, d2, 0x0C030201u);
                    }
                    tl4[t + sl * VIEWS_BLOCK] = o;
                }
        };
        // byte offsets of this thread's upper / lower tap pairs inside one LDS buffer; per pair only the
        // scalar (buffer base + 4 * joff) is added
        uint32_t tap_up[PXT], tap_lo[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            tap_up[j] = (MODE == 2 && list_tile) ? list_tap_up[j] : 4u * (uint32_t)tap[j];
            tap_lo[j] = (MODE == 2 && list_tile) ? list_tap_lo[j] : 4u * (uint32_t)(tap[j] + rowdw);
        }
        auto half = [&](int k, const Q16 (&qcur)[VIEWS_SLOTS], Q16 (&qnext)[VIEWS_SLOTS], uint4* tl4, uint32_t buf_bytes) {
            stage1(k, qcur, tl4);
            uint32_t soff = buf_bytes + 4u * (((uint32_t)__builtin_amdgcn_readlane((int)cw0, k) >> 20) & 3u);
            asm volatile("" : "+s"(soff));  // one scalar: keeps the buffer base out of four separate vector adds
            __syncthreads();
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
            if (k + 1 < nplain)
                load_pieces(k + 1, qnext);
            uint32_t pix[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                pix[j] = blend4_packed(ta[j][0], ta[j][1], ta[j][2], ta[j][3], tw[j]);
            store_pixels(pair0 + (int)((uint32_t)__builtin_amdgcn_readlane(cw3, k) >> 26), pix);
        };
        Q16 qa[VIEWS_SLOTS], qb[VIEWS_SLOTS];
        load_pieces(0, qa);
        for (int k = 0; k < nplain; k += 2) {
            half(k, qa, qb, tile4[0], 0u);
            if (k + 1 >= nplain)
                break;
            half(k + 1, qb, qa, tile4[1], (uint32_t)sizeof(tile4[0]));
        }
        if (nplain == npairs)
            return;
        __syncthreads();  // the last plain pair's taps are read before the general loop writes the buffers
    }

    PairCtx pc = pair_ctx(nplain);
    Q16 q[VIEWS_SLOTS];
    uint32_t fw[VIEWS_SLOTS];
    if (pc.fast)
        issue_loads(pc, src + (size_t)pc.pano * P.pano_stride, q, fw);

    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, st6 = 0;
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    (void)st0; (void)st1; (void)st2; (void)st3; (void)st4; (void)st5; (void)st6; (void)acc;
    int buf = 0;
    for (int ki = nplain; ki < npairs; ++ki) {
        STAMP(st0);
        const uint8_t* __restrict__ S = src + (size_t)pc.pano * P.pano_stride;
        const int cur_yaw = pc.yaw_i;
        const int pair = pair0 + pc.korig;
        const bool has_next = ki + 1 < npairs;
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
            const int joff = pc.joff;
            STAMP(st1);
            __syncthreads();
            STAMP(st2);
            // the 2x2 taps of this thread's pixels
            const uint32_t* tl = reinterpret_cast<const uint32_t*>(tl4);
            uint32_t ta[PXT][4];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                int b = tap[j] + joff, b2 = b + rowdw;
                if (MODE == 2 && list_tile) {
                    b = (int)(list_tap_up[j] >> 2) + joff;
                    b2 = (int)(list_tap_lo[j] >> 2) + joff;
                }
                ta[j][0] = tl[b];
                ta[j][1] = tl[b + 1];
                ta[j][2] = tl[b2];
                ta[j][3] = tl[b2 + 1];
            }
            STAMP(st3);
            // the next pair's source loads go out now; their latency hides behind stage 2
            if (has_next) {
                pc = pair_ctx(ki + 1);
                if (pc.fast)
                    issue_loads(pc, src + (size_t)pc.pano * P.pano_stride, q, fw);
            }
#ifdef P2P_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            STAMP(st4);
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                pix[j] = blend4_packed(ta[j][0], ta[j][1], ta[j][2], ta[j][3], tw[j]);
            STAMP(st5);
            buf ^= 1;  // the next pair writes the other buffer; its readers are past this barrier
        } else {
            direct_pixels(S, cur_yaw, pix);
            if (has_next) {
                pc = pair_ctx(ki + 1);
                if (pc.fast)
                    issue_loads(pc, src + (size_t)pc.pano * P.pano_stride, q, fw);
            }
        }
        store_pixels(pair, pix);
#ifdef P2P_STAMPS
        STAMP(st6);
        acc[0] += st1 - st0;
        acc[1] += st2 - st1;
        acc[2] += st3 - st2;
        acc[3] += st4 - st3;
        acc[4] += st5 - st4;
        acc[5] += st6 - st5;
#endif
    }
#ifdef P2P_STAMPS
    if ((t & 63) == 0) {
        const int slot = (int)(((uint32_t)bx * 7u + blockIdx.y * 131u + blockIdx.z * 977u + (t >> 6) * 1031u) & 4095u);
        for (int i = 0; i < 6; ++i)
            atomicAdd(&g_stamps[i * 4096 + slot], acc[i]);
        atomicAdd(&g_stamps[6 * 4096 + slot], 1ull);
        atomicAdd(&g_stamps[7 * 4096 + slot], (unsigned long long)(pair1 - pair0));
    }
#endif
}

// MODE 0 launches carry the sub-tile workgroups in front of the tile workgroups (blockIdx.x < P.plan_gx):
// one launch, so the few long-running sub-tile passes overlap with the bulk instead of trailing it.
template <int MAPSRC, int MODE>
__global__ __launch_bounds__(VIEWS_BLOCK, 6) void remap_views_kernel(
    ViewsParams P, const uint8_t* __restrict__ src, const uint32_t* __restrict__ ytab,
    const YawDesc* __restrict__ ydesc, const uint32_t* __restrict__ f4tab,
    const PitchConst* __restrict__ pitch, const float* __restrict__ mapU,
    const float* __restrict__ mapV, uint8_t* __restrict__ out, int32_t* __restrict__ coords)
{
    __shared__ uint4 tile4[2][LDS_ITEMS_CAP];
    __shared__ int bbox[2 * VIEWS_BLOCK / 64];
    __shared__ uint32_t half_box[2][4];
    if (MODE == 0 && (int)blockIdx.x < P.plan_gx) {
        views_body<MAPSRC, 2>(P, src, ytab, ydesc, f4tab, pitch, mapU, mapV, out, coords, tile4, bbox, half_box,
                              (int)blockIdx.x, P.plan_gx);
        return;
    }
    views_body<MAPSRC, MODE>(P, src, ytab, ydesc, f4tab, pitch, mapU, mapV, out, coords, tile4, bbox, half_box,
                             (int)blockIdx.x - (MODE == 0 ? P.plan_gx : 0), (int)gridDim.x - (MODE == 0 ? P.plan_gx : 0));
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int MODE>
static void launch_views_mode(const ViewsParams& P, int mapsrc, dim3 grid, hipStream_t st)
{
    if (mapsrc == 1)
        hipLaunchKernelGGL((remap_views_kernel<1, MODE>), grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ytab, P.ydesc,
                           P.f4tab, P.pitch, P.mapU, P.mapV, P.out, P.coords);
    else if (mapsrc == 2)
        hipLaunchKernelGGL((remap_views_kernel<2, MODE>), grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ytab, P.ydesc,
                           P.f4tab, P.pitch, P.mapU, P.mapV, P.out, P.coords);
    else
        hipLaunchKernelGGL((remap_views_kernel<0, MODE>), grid, dim3(VIEWS_BLOCK), 0, st, P, P.src, P.ytab, P.ydesc,
                           P.f4tab, P.pitch, P.mapU, P.mapV, P.out, P.coords);
}

// mode 0: every tile of every view, preceded by the P.plan_n listed sub-tiles; mode 1: the plan pass (one
// workgroup per tile and pitch, nothing drawn)
hipError_t launch_remap_views(const ViewsParams& P, int mapsrc, int mode, hipStream_t st)
{
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const int n_pairs = P.n_panos * P.n_yaw;
    const int zblocks = (n_pairs + P.pairs_per_block - 1) / P.pairs_per_block;
    if (mode == 1)
        launch_views_mode<1>(P, mapsrc, dim3(8 * ((tiles + 7) / 8), P.n_pitch, 1), st);
    else  // 8 XCDs, each a contiguous run of tiles; P.plan_gx is a multiple of 8 too
        launch_views_mode<0>(P, mapsrc, dim3(P.plan_gx + 8 * ((tiles + 7) / 8), P.n_pitch, zblocks), st);
    return hipGetLastError();
}


hipError_t read_stamps(unsigned long long* out16, bool reset)
{
    static unsigned long long host[8 * 4096];
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(host));
    for (int i = 0; i < 16; ++i)
        out16[i] = 0;
    for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 4096; ++k)
            out16[i] += host[i * 4096 + k];
    if (e == hipSuccess && reset) {
        for (auto& v : host)
            v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), host, sizeof(host));
    }
    return e;
}

}  // namespace p2p
