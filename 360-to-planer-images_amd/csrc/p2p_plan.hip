// p2p_plan.hip -- the plan pass of the view kernel: everything that depends on the coordinate maps only
//   get_pitch_mapping / pitch_mapping_cache  P:17-18, P:55-73  -> plan_kernel (once per job geometry)
// Reference behaviour (cited, never copied):
//   P = /root/reference/app/panorama_to_plane-pitch.py
// The reference evaluates each pitch map once and keeps it for the life of the process (pitch_mapping_cache,
// key (ow, oh, pitch, pw, ph, fov)); every later yaw and every later image re-uses it.  The device analogue:
// one pass per job geometry evaluates (or reads, for caller maps) the float32 map of every output pixel,
// quantises it as cv::remap does (INTER_BITS = 5) and stores
//   * the quantised coordinates (sx, sy), for the direct-gather path and for p2p_job_get_coords;
//   * per tile: the footprint of the tile in the yaw-resampled panorama as per-row spans of 4-pixel items, and
//     per pixel the LDS offsets of its 2x2 taps inside that footprint plus its two 5-bit weights (one dword);
//   * tiles whose footprint does not fit the LDS buffers (strong minification, a pole inside) or touches the
//     panorama's border are marked for the gather kernel and entered in a list.
// No pixel data is touched here.  The view kernel (p2p_views.hip) then starts every launch from these tables.
// Compiled with -ffp-contract=off: every float operation rounds where NumPy rounds.
#include "p2p_inline.h"

namespace p2p {
namespace P2P_SHAPE_NS {

namespace {

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, total in *total
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* s_scan, uint32_t* total)
{
    const int t = threadIdx.x;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if ((t & 63) >= d)
            incl += up;
    }
    if ((t & 63) == 63)
        s_scan[t >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - v, all = 0u;
#pragma unroll
    for (int w = 0; w < VIEWS_BLOCK / 64; ++w) {
        base += w < (t >> 6) ? s_scan[w] : 0u;
        all += s_scan[w];
    }
    *total = all;
    return base;
}

// min / max over the 64 lanes of a wave, in every lane: a Hillis-Steele scan inside each row of 16 lanes (DPP
// row_shr), the rows' totals passed on by row_bcast, the wave's total read from lane 63.  Lanes that a step gives
// no source keep their value.
template <bool MAX>
__device__ __forceinline__ int wave_reduce(int v)
{
#define P2P_DPP_STEP(ctrl, row_mask)                                                  \
    {                                                                                 \
        const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, row_mask, 0xf, false);  \
        v = MAX ? max(v, o) : min(v, o);                                              \
    }
    P2P_DPP_STEP(0x111, 0xf)  // row_shr:1
    P2P_DPP_STEP(0x112, 0xf)  // row_shr:2
    P2P_DPP_STEP(0x114, 0xf)  // row_shr:4
    P2P_DPP_STEP(0x118, 0xf)  // row_shr:8
    P2P_DPP_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1 and 3
    P2P_DPP_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2 and 3
#undef P2P_DPP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

}  // namespace

template <bool CALLER_MAPS>
__global__ __launch_bounds__(VIEWS_BLOCK) void plan_kernel(PlanParams P)
{
    __shared__ int4 s_wbox[VIEWS_BLOCK / 64];   // per wave: min ix, max ix, min iy, max iy of its live pixels
    __shared__ int s_wflags[VIEWS_BLOCK / 64];  // per wave: bit 0 any live pixel, bit 1 any pixel outside the panorama under a non-constant border
    __shared__ int s_rmin[PLAN_MAX_ROWS], s_rmax[PLAN_MAX_ROWS];
    __shared__ uint32_t s_rbase[PLAN_MAX_ROWS + 1];
    __shared__ uint32_t s_scan[VIEWS_BLOCK / 64];

    constexpr int PXT = VIEWS_PXT;
    constexpr int ROWSTEP = VIEWS_BLOCK / TILE_W;
    const int t = threadIdx.x;
    const int tiles_x = (P.ow + TILE_W - 1) / TILE_W;
    const int tiles_y = (P.oh + TILE_H - 1) / TILE_H;
    const int tile_id = blockIdx.x, pitch_i = blockIdx.y;
    const int x0 = (tile_id % tiles_x) * TILE_W, y0 = (tile_id / tiles_x) * TILE_H;
    const int tx = t % TILE_W, ty0 = t / TILE_W;
    const int px = x0 + tx;
    const uint32_t slot = (uint32_t)(pitch_i * tiles_x * tiles_y + tile_id);

    // ---- the pitch-stage coordinate of every pixel of the tile, quantised as cv::remap does ----
    int ix[PXT], iy[PXT];
    uint32_t fx[PXT], fy[PXT], frac16[PXT];
    bool inside[PXT], inrange[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const int py = y0 + ty0 + j * ROWSTEP;
        inside[j] = px < P.ow && py < P.oh;
        frac16[j] = 0u;
        if (P.float_path) {
            // Float pixel path (not in the reference): one float resample at (U + yaw shift, V), true wrap-around in
            // U.  The plan keeps floor(U), floor(V) (as tap offsets) and the fractions in 1/65536; the yaw's
            // fractional shift is added per yaw by the kernel and may carry into the next column, so the row spans
            // are one column wider.  The bottom row is folded onto (ph - 2, fraction 1) so that the lower tap exists.
            ix[j] = iy[j] = -32768;
            fx[j] = fy[j] = 0u;
            inrange[j] = false;
            if (inside[j]) {
                const size_t k = ((size_t)pitch_i * P.oh + py) * P.ow + px;
                const PitchConst pc = P.pitch[pitch_i];
                float U, V;
                pitch_map_eval((float)px + P.centre, (float)py + P.centre, P.geom, pc.c, pc.s, U, V, false);
                float uc = U - P.centre, vc = V - P.centre;
                if (uc < 0.0f)
                    uc += P.geom.pw_f;
                if (vc < 0.0f)
                    vc = 0.0f;
                const bool dead = !(U == U) || !(V == V);  // NaN next to a pole: black, as in the exact path
                P.coords[k] = make_int2(__float_as_int(dead ? __int_as_float(0x7FC00000) : uc), __float_as_int(vc));
                if (!dead && P.ph >= 2) {
                    int xi = (int)uc;
                    if (xi >= P.pw)
                        xi = P.pw - 1;
                    float fu = uc - (float)xi, fv;
                    int yi = (int)vc;
                    if (yi >= P.ph - 1) {
                        yi = P.ph - 2;
                        fv = 1.0f;
                    } else {
                        fv = vc - (float)yi;
                    }
                    const uint32_t fu16 = min(65535u, (uint32_t)(fu * 65536.0f)), fv16 = min(65535u, (uint32_t)(fv * 65536.0f));
                    frac16[j] = fu16 | fv16 << 16;
                    ix[j] = xi;
                    iy[j] = yi;
                    inrange[j] = true;
                }
            }
            continue;
        }
        int sx = INT32_MIN, sy = INT32_MIN;
        if (inside[j]) {
            const size_t k = ((size_t)pitch_i * P.oh + py) * P.ow + px;
            float U, V;
            if (CALLER_MAPS) {
                U = P.mapU[k];
                V = P.mapV[k];
            } else {
                const PitchConst pc = P.pitch[pitch_i];
                pitch_map_eval((float)px, (float)py, P.geom, pc.c, pc.s, U, V);
            }
            sx = cv_round_f32(U * 32.0f);
            sy = cv_round_f32(V * 32.0f);
            P.coords[k] = make_int2(sx, sy);
        }
        ix[j] = sat_short(sx >> 5);
        iy[j] = sat_short(sy >> 5);
        fx[j] = (uint32_t)sx & 31u;
        fy[j] = (uint32_t)sy & 31u;
        // A pixel contributes only if its 2x2 footprint touches the panorama (BORDER_CONSTANT 0: cv::remap writes
        // borderValue when sx >= w || sx+1 < 0 || sy >= h || sy+1 < 0).  For the reference's clipped maps that is
        // every pixel except NaN ones (ix = iy = -32768).
        inrange[j] = inside[j] && ix[j] >= -1 && iy[j] >= -1 && ix[j] < P.pw && iy[j] < P.ph;
    }

    // ---- bounding box of the live pixels' coordinates: per thread, per wave (DPP), then over the waves' four entries.
    // (An atomicMin on ONE LDS word per pixel is turned into a scalar loop over the wave's lanes by the compiler:
    // sixteen of them were a thousand v_readlane per wave, two thirds of this kernel's instructions.)
    int bx0 = INT32_MAX, bx1 = INT32_MIN, by0 = INT32_MAX, by1 = INT32_MIN;
    bool live = false, strayp = false;
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        if (inrange[j]) {
            bx0 = min(bx0, ix[j]); bx1 = max(bx1, ix[j]);
            by0 = min(by0, iy[j]); by1 = max(by1, iy[j]);
            live = true;
        } else if (inside[j] && P.border != 0) {
            strayp = true;  // reads reflected / wrapped / replicated pixels: cv::borderInterpolate, table path
        }
    }
    // how many source rows one output row of 64 pixels runs across (the widest of the wave's rows)
    int across = 0;
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const int lo = wave_reduce<false>(inrange[j] ? iy[j] : INT32_MAX), hi = wave_reduce<true>(inrange[j] ? iy[j] : INT32_MIN);
        across = max(across, hi >= lo ? hi - lo : 0);
    }
    bx0 = wave_reduce<false>(bx0); bx1 = wave_reduce<true>(bx1);
    by0 = wave_reduce<false>(by0); by1 = wave_reduce<true>(by1);
    const bool wlive = __ballot(live) != 0ull, wstray = __ballot(strayp) != 0ull;
    if ((t & 63) == 0) {
        s_wbox[t >> 6] = make_int4(bx0, bx1, by0, by1);
        s_wflags[t >> 6] = (int)wlive | (int)wstray << 1 | (across >= P.blocky_from ? 4 : 0);
    }
    __syncthreads();
    int c0 = INT32_MAX, c1 = INT32_MIN, r0 = INT32_MAX, r1 = INT32_MIN, fl = 0;
#pragma unroll
    for (int w = 0; w < VIEWS_BLOCK / 64; ++w) {
        const int4 b = s_wbox[w];
        c0 = min(c0, b.x); c1 = max(c1, b.y); r0 = min(r0, b.z); r1 = max(r1, b.w);
        fl |= s_wflags[w];
    }
    const bool any_live = (fl & 1) != 0, stray = (fl & 2) != 0;
    const int nrow = any_live ? r1 - r0 + 2 : 0;
    // The LDS scheme needs the whole footprint strictly inside the panorama (no tap is a border tap) and a
    // panorama width divisible by 4 (12-byte items never straddle a row end).
    // (float path: taps reach one column further)
    bool ok = any_live && !stray && (P.pw & 3) == 0 && c0 >= 0 && r0 >= 0 && c1 + 1 + P.float_path < P.pw &&
              r1 + 1 < P.ph && nrow <= PLAN_MAX_ROWS && nrow <= VIEWS_BLOCK;  // (one plan thread per rot row)
    uint32_t n_items = 0;
    if (ok) {
        // ---- per rot row: the span of columns the taps read.  A pixel's taps sit in rows iy and iy + 1: two atomics
        // per pixel on its own row, then the rows' threads join row r with row r - 1 ----
        if (t < nrow) {
            s_rmin[t] = INT32_MAX;
            s_rmax[t] = -1;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PXT; ++j)
            if (inrange[j]) {
                const int r = iy[j] - r0;
                atomicMin(&s_rmin[r], ix[j]);
                atomicMax(&s_rmax[r], ix[j] + 1 + P.float_path);
            }
        __syncthreads();
        int jmin = INT32_MAX, jmax = -1;
        if (t < nrow) {
            jmin = s_rmin[t];
            jmax = s_rmax[t];
            if (t > 0) {
                jmin = min(jmin, s_rmin[t - 1]);
                jmax = max(jmax, s_rmax[t - 1]);
            }
        }
        __syncthreads();
        if (t < nrow) {
            s_rmin[t] = jmin;
            s_rmax[t] = jmax;
        }
        // a row's LDS span starts at a column congruent to c0 mod 4 (so that one per-yaw alignment serves
        // every row) and leaves room for the yaw's alignment 0..3 within the first item
        int o = 0;
        uint32_t width = 0;
        if (t < nrow && s_rmax[t] >= 0) {
            o = s_rmin[t] - ((s_rmin[t] - c0) & 3);
            width = (uint32_t)(((3 + s_rmax[t] - o) >> 2) + 1);
        }
        const uint32_t base = block_scan_excl(width, s_scan, &n_items);
        ok = n_items <= (uint32_t)LDS_ITEMS_CAP;
        if (ok && t < nrow) {
            s_rmin[t] = o;
            s_rbase[t] = base;
            s_rmax[t] = (int)width;
        }
        __syncthreads();
        if (ok) {
            // the per-pixel word keeps (lower tap - upper tap) in PXW_DL_BITS bits: the distance between the
            // same column in two consecutive rows (about one row's width) -- a tile with a longer row gathers
            bool far = false;
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                if (inrange[j]) {
                    const int r = iy[j] - r0;
                    const uint32_t up = 4u * s_rbase[r] + (uint32_t)(ix[j] - s_rmin[r]);
                    const uint32_t lo = 4u * s_rbase[r + 1] + (uint32_t)(ix[j] - s_rmin[r + 1]);
                    far |= lo - up >= (1u << PXW_DL_BITS);
                }
            ok = __syncthreads_or(far) == 0;
        }
    }
    uint32_t* pxw = P.px + (size_t)slot * (VIEWS_BLOCK * PXT);
    uint32_t* px2w = P.float_path ? P.px2 + (size_t)slot * (VIEWS_BLOCK * PXT) : nullptr;
    uint32_t* itw = P.items + (size_t)slot * LDS_ITEMS_CAP;
    // per-pixel words in the tile's thread order (thread t draws pixels j = 0..PXT-1, rows ROWSTEP apart); pixels
    // outside the view, pixels without a footprint and the pixels of a gather tile read as zero; so do unused items
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        uint32_t word = 0u;
        if (ok && inrange[j]) {
            const int r = iy[j] - r0;
            const uint32_t up = 4u * s_rbase[r] + (uint32_t)(ix[j] - s_rmin[r]);
            const uint32_t lo = 4u * s_rbase[r + 1] + (uint32_t)(ix[j] - s_rmin[r + 1]);
            word = up | (lo - up) << PXW_UP_BITS | fx[j] << 22 | fy[j] << 27;
        }
        pxw[j * VIEWS_BLOCK + t] = word;
        if (px2w)
            px2w[j * VIEWS_BLOCK + t] = ok ? frac16[j] : 0u;
    }
    if (ok && t < nrow) {
        const uint32_t g0 = (uint32_t)((s_rmin[t] - c0) >> 2);
        for (int g = 0; g < s_rmax[t]; ++g)
            itw[s_rbase[t] + g] = (uint32_t)(r0 + t) << 16 | (g0 + (uint32_t)g);
    }
    for (uint32_t k = (ok ? n_items : 0u) + (uint32_t)t; k < (uint32_t)LDS_ITEMS_CAP; k += VIEWS_BLOCK)
        itw[k] = 0u;
    if (t == 0) {
        if (!ok)  // tiles for the gather kernels, listed
            P.gather_list[atomicAdd(P.n_gather, 1u)] = slot;
        PieceHdr h;
        h.mode_items = (ok ? (1u | n_items << 8) : 2u) | (uint32_t)(fl & 4);
        h.c0 = any_live ? c0 : 0;
        h.c1 = any_live ? c1 : -1;
        // rot rows of the live pixels' upper taps, + 1 (16 bits each): the host orders the gather list by source position
        h.rows = any_live ? ((uint32_t)(r0 + 1) & 0xFFFFu) | (uint32_t)(r1 + 1) << 16 : 0u;
        P.hdr[slot] = h;
    }
}

hipError_t launch_plan(const PlanParams& P, hipStream_t st)
{
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    if (P.mapU)
        hipLaunchKernelGGL(plan_kernel<true>, dim3(tiles, P.n_pitch), dim3(VIEWS_BLOCK), 0, st, P);
    else
        hipLaunchKernelGGL(plan_kernel<false>, dim3(tiles, P.n_pitch), dim3(VIEWS_BLOCK), 0, st, P);
    return hipGetLastError();
}

}  // namespace P2P_SHAPE_NS

// what the host calls this shape through (p2p_device.h: tile shapes)
const ShapeOps& P2P_SHAPE_OPS_NAME()
{
    static const ShapeOps ops = {{P2P_SHAPE_NS::TILE_W, P2P_SHAPE_NS::TILE_H, P2P_SHAPE_NS::VIEWS_BLOCK, P2P_SHAPE_NS::VIEWS_PXT, P2P_SHAPE_NS::LDS_ITEMS_CAP},
                                 &P2P_SHAPE_NS::launch_plan, &P2P_SHAPE_NS::launch_remap_views, &P2P_SHAPE_NS::launch_float_views};
    return ops;
}

}  // namespace p2p
