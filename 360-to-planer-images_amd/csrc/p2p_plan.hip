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
// no source keep their value: the DPP move's `old` operand is the operation's identity, so that min / max with it changes
// nothing -- and so that the compiler folds the move into the operation (v_min_i32_dpp; with `old` = the value itself it
// emitted v_mov + v_mov_dpp + v_min + s_nop per step).
template <bool MAX>
__device__ __forceinline__ int wave_reduce(int v)
{
#define P2P_DPP_STEP(ctrl, row_mask)                                                  \
    {                                                                                 \
        const int o = __builtin_amdgcn_update_dpp(MAX ? INT32_MIN : INT32_MAX, v, ctrl, row_mask, 0xf, false);  \
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

constexpr int PLAN_CELL_SLOTS_LOG2 = 7, PLAN_CELL_SLOTS = 1 << PLAN_CELL_SLOTS_LOG2, PLAN_CELL_PROBES = 16;  // (band plans: the tile's cell table)

// A tile for the gather kernels goes into the plan's list.  The counter is the context's (zero between plan passes); were it
// ever not -- this pool has lost writes to fresh device memory under allocation churn, DESIGN.md section 5.7 -- the entry is
// dropped rather than written past the list, and the host sees a count it cannot believe.
__device__ __forceinline__ void list_gather_tile(const PlanParams& P, uint32_t slot)
{
    const uint32_t at = atomicAdd(P.n_gather, 1u);
    const uint32_t slots = (uint32_t)P.n_pitch * (uint32_t)(((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H));
    if (at < slots)
        P.gather_list[at] = slot;
}

template <bool CALLER_MAPS>
__device__ __forceinline__ void plan_tile(const PlanParams& P)
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

    // a tile row the job does not draw (p2p_job_set_rows: one image's rows shared out to several GPUs): mode 0, no kernel's
    if (tile_id / tiles_x < P.ty0 || tile_id / tiles_x >= P.ty1) {
        if (P.band.gcell != nullptr && (t & 3) == 0) {
            const size_t gxn = (size_t)((P.ow + 3) >> 2);
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const int py = y0 + ty0 + j * ROWSTEP;
                if (px < P.ow && py < P.oh)
                    P.band.gcell[((size_t)pitch_i * P.oh + py) * gxn + (size_t)(px >> 2)] = ~0u;
            }
        }
        if (t == 0) {
            PieceHdr h;
            h.mode_items = 0u;
            h.c0 = 0;
            h.c1 = -1;
            h.rows = 0u;
            P.hdr[slot] = h;
            if (P.hdr_host != nullptr)
                P.hdr_host[slot] = h;
        }
        return;
    }

    // ---- the pitch-stage coordinate of every pixel of the tile, quantised as cv::remap does ----
    int ix[PXT], iy[PXT], qsx[PXT], qsy[PXT];
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
            qsx[j] = qsy[j] = INT32_MIN;
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
            if (P.coords_all)
                P.coords[k] = make_int2(sx, sy);
        }
        qsx[j] = sx;
        qsy[j] = sy;
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
    // ---- band plan: the tile's 4-pixel groups (lanes 4k .. 4k + 3 of a wave are four neighbours of one row) go into
    // source-band tiles if every one of them can: some live pixel, no tap on the panorama's border, taps within
    // maxw x maxh of each other.  One bad group sends the whole tile to the gather kernel (as a pole does above).
    const bool band = P.band.gcell != nullptr;
    int gq_x0[PXT], gq_y0[PXT], gq_x1[PXT], gq_y1[PXT];
    bool band_ok = false;
    if (band) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const int py = y0 + ty0 + j * ROWSTEP;
            const bool exists = (px & ~3) < P.ow && py < P.oh;
            int qx0 = inrange[j] ? ix[j] : INT32_MAX, qx1 = inrange[j] ? ix[j] : INT32_MIN;
            int qy0 = inrange[j] ? iy[j] : INT32_MAX, qy1 = inrange[j] ? iy[j] : INT32_MIN;
// (every lane of a quad has a source; the identity as the move's `old` value lets the compiler fold move and operation)
#define P2P_QUAD_STEP(ctrl)                                                                          \
            qx0 = min(qx0, __builtin_amdgcn_update_dpp(INT32_MAX, qx0, ctrl, 0xf, 0xf, false));      \
            qx1 = max(qx1, __builtin_amdgcn_update_dpp(INT32_MIN, qx1, ctrl, 0xf, 0xf, false));      \
            qy0 = min(qy0, __builtin_amdgcn_update_dpp(INT32_MAX, qy0, ctrl, 0xf, 0xf, false));      \
            qy1 = max(qy1, __builtin_amdgcn_update_dpp(INT32_MIN, qy1, ctrl, 0xf, 0xf, false));
            P2P_QUAD_STEP(0xB1)  // quad_perm:[1,0,3,2]
            P2P_QUAD_STEP(0x4E)  // quad_perm:[2,3,0,1]
#undef P2P_QUAD_STEP
            gq_x0[j] = qx0; gq_x1[j] = qx1; gq_y0[j] = qy0; gq_y1[j] = qy1;
            const bool edge = inrange[j] && (ix[j] < 0 || ix[j] + 1 >= P.pw || iy[j] < 0 || iy[j] + 1 >= P.ph);
            const bool dead = exists && qx1 < qx0;
            const bool wide = exists && qx1 >= qx0 && (qx1 - qx0 > P.band.g.maxw || qy1 - qy0 > P.band.g.maxh);
            bad |= edge || dead || wide;
        }
        // A tile's 256 groups fall into a few dozen cells: they are counted in a small LDS table first (keyed by cell,
        // linear probing) and reach the global cells as one set of atomics per (tile, cell) instead of one per group --
        // 3.2 M global atomics on the reference CLI's default set otherwise.  A group that finds no slot goes straight
        // to the global cells.
        __shared__ uint32_t s_ckey[PLAN_CELL_SLOTS], s_ccnt[PLAN_CELL_SLOTS];
        __shared__ int s_cminv[PLAN_CELL_SLOTS], s_cmax1[PLAN_CELL_SLOTS], s_crmax1[PLAN_CELL_SLOTS];
        if (t < PLAN_CELL_SLOTS) {
            s_ckey[t] = ~0u; s_ccnt[t] = 0u; s_cminv[t] = 0; s_cmax1[t] = 0; s_crmax1[t] = 0;
        }
        band_ok = __syncthreads_or(bad) == 0 && any_live && !stray && (P.pw & 3) == 0 && !P.float_path;
        // its groups into the cells of the source (or marked: the tile gathers)
        if ((t & 3) == 0) {
            const size_t gxn = (size_t)((P.ow + 3) >> 2);
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const int py = y0 + ty0 + j * ROWSTEP;
                if (px < P.ow && py < P.oh) {
                    uint32_t cell = ~0u;
                    if (band_ok) {
                        cell = (uint32_t)(gq_y0[j] / P.band.g.bh) * (uint32_t)P.band.g.ncx + (uint32_t)(gq_x0[j] / P.band.g.cw);
                        uint32_t h = (cell * 2654435761u) >> (32 - PLAN_CELL_SLOTS_LOG2);
                        bool found = false;
                        for (int probe = 0; probe < PLAN_CELL_PROBES && !found; ++probe) {
                            const uint32_t prev = atomicCAS(&s_ckey[h], ~0u, cell);
                            found = prev == ~0u || prev == cell;
                            if (!found)
                                h = (h + 1u) & (uint32_t)(PLAN_CELL_SLOTS - 1);
                        }
                        if (found) {
                            atomicAdd(&s_ccnt[h], 1u);
                            atomicMax(&s_cminv[h], INT32_MAX - gq_x0[j]);
                            atomicMax(&s_cmax1[h], gq_x1[j] + 1);
                            atomicMax(&s_crmax1[h], gq_y1[j] + 1);
                        } else {
                            atomicAdd(&P.band.cell_count[cell], 1u);
                            atomicMax(&P.band.cell_cmin[cell], INT32_MAX - gq_x0[j]);  // (the minimum, kept inverted: the host zeroes the cells with one memset)
                            atomicMax(&P.band.cell_cmax1[cell], gq_x1[j] + 1);
                            atomicMax(&P.band.cell_rmax1[cell], gq_y1[j] + 1);
                        }
                    }
                    P.band.gcell[((size_t)pitch_i * P.oh + py) * gxn + (size_t)(px >> 2)] = cell;
                }
            }
        }
        __syncthreads();
        if (t < PLAN_CELL_SLOTS && s_ckey[t] != ~0u) {
            const uint32_t cell = s_ckey[t];
            atomicAdd(&P.band.cell_count[cell], s_ccnt[t]);
            atomicMax(&P.band.cell_cmin[cell], s_cminv[t]);
            atomicMax(&P.band.cell_cmax1[cell], s_cmax1[t]);
            atomicMax(&P.band.cell_rmax1[cell], s_crmax1[t]);
        }
    }
    // The LDS scheme needs the whole footprint strictly inside the panorama (no tap is a border tap) and a
    // panorama width divisible by 4 (12-byte items never straddle a row end).
    // (float path: taps reach one column further)
    bool ok = !band && any_live && !stray && (P.pw & 3) == 0 && c0 >= 0 && r0 >= 0 && c1 + 1 + P.float_path < P.pw &&
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
    if (band) {  // no tables: mode 3 tiles are drawn from the band tiles, the others from the coordinates
        if (t == 0) {
            if (!band_ok)
                list_gather_tile(P, slot);
            PieceHdr h;
            h.mode_items = (band_ok ? 3u : 2u) | (uint32_t)(fl & 4);
            h.c0 = any_live ? c0 : 0;
            h.c1 = any_live ? c1 : -1;
            h.rows = any_live ? ((uint32_t)(r0 + 1) & 0xFFFFu) | (uint32_t)(r1 + 1) << 16 : 0u;
            P.hdr[slot] = h;
            if (P.hdr_host != nullptr)  // (the host makes the gather tiles' lists from the headers while the band tiles are built)
                P.hdr_host[slot] = h;
        }
        return;
    }
    // The quantised coordinates are what the gather and table kernels draw from: kept for the tiles that are theirs (and
    // for all tiles when the host asks: band plans, the float path, p2p_job_get_coords -- coords_kernel fills them in
    // later otherwise).  A plan none of whose tiles gathers writes none: 8 bytes per output pixel, config 2's plan pass
    // wrote 100 MB of them next to 42 MB of tables.
    if (!P.coords_all && !ok && !P.float_path) {
#pragma unroll
        for (int j = 0; j < PXT; ++j)
            if (inside[j])
                P.coords[((size_t)pitch_i * P.oh + (y0 + ty0 + j * ROWSTEP)) * P.ow + px] = make_int2(qsx[j], qsy[j]);
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
            list_gather_tile(P, slot);
        PieceHdr h;
        h.mode_items = (ok ? (1u | n_items << 8) : 2u) | (uint32_t)(fl & 4);
        h.c0 = any_live ? c0 : 0;
        h.c1 = any_live ? c1 : -1;
        // rot rows of the live pixels' upper taps, + 1 (16 bits each): the host orders the gather list by source position
        h.rows = any_live ? ((uint32_t)(r0 + 1) & 0xFFFFu) | (uint32_t)(r1 + 1) << 16 : 0u;
        P.hdr[slot] = h;
    }
}


// ---------------------------------------------------------------------------------------------
// Source-band tiles (p2p_device.h: BandGeom ...).  The plan pass above has counted every group of every mode-3 tile into
// the cell of its upper-left tap (cell_count, and the cell's tap extents) and written each group's cell (gcell).
//   band_cut_kernel      one wave per band: walks the band's cells from left to right and cuts the row into tiles --
//                        a tile takes cells (the last one possibly in part) until it holds VIEWS_BLOCK groups or one
//                        more cell would push its rectangle (first cell's leftmost tap .. rightmost tap, band top ..
//                        lowest tap) beyond LDS_ITEMS_CAP items.  Run twice: count, then (bases known) write.
//   band_scan_kernel     one workgroup: exclusive scan of the bands' tile and group counts.
//   band_scatter_kernel  every group to its cell's run of the sorted list (order inside a cell: as the atomics fall).
//   band_build_kernel    one workgroup per tile: its groups sorted by index = (pitch view, row, column), one per lane;
//                        the exact rectangle of their taps, the per-pixel words, the group words, the header.
//   band_xcd_kernel      one workgroup: the tiles (band order = source order) cut into eight runs of equal work.
// ---------------------------------------------------------------------------------------------
constexpr int BAND_MAX_NCX = 4096;  // cells per band the cut keeps in LDS (64 KB): panoramas up to 32766 wide in cells of >= 8 columns

// inclusive scans over the 64 lanes of a wave
// (DPP: Hillis-Steele inside each row of 16 lanes, then the rows' totals handed on -- a ds_bpermute per step would cost
// an LDS round trip each, and the cut is a chain of such scans.)  Values are >= 0: lanes a step gives no source add / max 0.
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v, int)
{
#define P2P_DPP_ADD(ctrl, row_mask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, row_mask, 0xf, false);
    P2P_DPP_ADD(0x111, 0xf) P2P_DPP_ADD(0x112, 0xf) P2P_DPP_ADD(0x114, 0xf) P2P_DPP_ADD(0x118, 0xf)
    P2P_DPP_ADD(0x142, 0xa) P2P_DPP_ADD(0x143, 0xc)
#undef P2P_DPP_ADD
    return v;
}
__device__ __forceinline__ int wave_scan_max(int v, int)
{
#define P2P_DPP_MAX(ctrl, row_mask) v = max(v, __builtin_amdgcn_update_dpp(0, v, ctrl, row_mask, 0xf, false));
    P2P_DPP_MAX(0x111, 0xf) P2P_DPP_MAX(0x112, 0xf) P2P_DPP_MAX(0x114, 0xf) P2P_DPP_MAX(0x118, 0xf)
    P2P_DPP_MAX(0x142, 0xa) P2P_DPP_MAX(0x143, 0xc)
#undef P2P_DPP_MAX
    return v;
}

template <bool WRITE>
__global__ __launch_bounds__(64) void band_cut_kernel(BandParams B)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int r0 = b * B.g.bh;
    const size_t cbase = (size_t)b * B.g.ncx;
    const uint32_t tbase = WRITE ? B.band_tiles[b] : 0u, gbase = WRITE ? B.band_groups[b] : 0u;
    // the open tile (carried from window to window): its groups, first column, rightmost column, lowest row
    uint32_t ng = 0u, tiles = 0u, placed = 0u, tile_gstart = 0u;
    unsigned long long cost = 0ull;
    int cs = 0, cm = 0, rm = 0;
    auto close = [&]() {
        if (WRITE && lane == 0) {
            BandTileRec r;
            r.gstart = gbase + tile_gstart; r.gcount = ng; r.c0 = cs; r.cmax1 = cm; r.r0 = r0; r.rmax1 = rm;
            B.recs[tbase + tiles] = r;
        }
        cost += B.cost_base + (uint32_t)((((3 + cm - cs) >> 2) + 1) * (rm - r0 + 1));
        ++tiles;
        tile_gstart += ng;
        ng = 0u;
    };
    // One window of 64 cells at a time; inside it one step per TILE (not per cell): prefix sums / maxima from the
    // tile's first cell say in which lane the tile ends -- the first whose cell would push the rectangle beyond the LDS
    // buffer (the tile ends before it), or the first in which the groups reach VIEWS_BLOCK (the tile takes part of it).
    // (the band's cells into LDS first, all loads in flight at once: the walk itself is a chain of short steps)
    __shared__ uint4 s_cell[BAND_MAX_NCX];
    for (int ci = lane; ci < B.g.ncx && ci < BAND_MAX_NCX; ci += 64) {
        uint4 v;
        v.x = B.cell_count[cbase + ci];
        v.y = (uint32_t)(INT32_MAX - B.cell_cmin[cbase + ci]);
        v.z = (uint32_t)B.cell_cmax1[cbase + ci];
        v.w = (uint32_t)B.cell_rmax1[cbase + ci];
        s_cell[ci] = v;
    }
    __syncthreads();
    for (int c0i = 0; c0i < B.g.ncx; c0i += 64) {
        const int ci = c0i + lane;
        const uint4 cv = ci < B.g.ncx && ci < BAND_MAX_NCX ? s_cell[ci] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t n = cv.x;
        const int a = (int)cv.y, mx = (int)cv.z, rx = (int)cv.w;
        const uint32_t incl = wave_scan_add(n, lane);
        if (WRITE && ci < B.g.ncx)
            B.cell_off[cbase + ci] = gbase + placed + incl - n;
        placed += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        int at = 0;              // first lane not consumed yet
        uint32_t left = 0u;      // groups of lane `at` still to place (0: all of them)
        for (;;) {
            const uint32_t nn = lane < at ? 0u : (lane == at && left ? left : n);
            const unsigned long long nz = __ballot(nn != 0u);
            if (!nz)
                break;  // window exhausted; an open tile is carried on
            const int f = __ffsll((long long)nz) - 1;
            const int cs_t = ng ? cs : __builtin_amdgcn_readlane(a, f);
            const uint32_t C = wave_scan_add(nn, lane) + ng;
            const int M = max(wave_scan_max(nn ? mx : 0, lane), ng ? cm : 0);
            const int R = max(wave_scan_max(nn ? rx : 0, lane), ng ? rm : 0);
            const int items = (((3 + M - cs_t) >> 2) + 1) * (R - r0 + 1);
            const unsigned long long over = __ballot(nn != 0u && items > LDS_ITEMS_CAP);
            const unsigned long long full = __ballot(nn != 0u && C >= (uint32_t)VIEWS_BLOCK);
            const int e_cap = over ? __ffsll((long long)over) - 1 : 64, e_cnt = full ? __ffsll((long long)full) - 1 : 64;
            if (e_cap <= e_cnt && e_cap < 64) {
                // the cell of lane e_cap does not fit: the tile ends before it (it holds something: one cell always fits)
                if (e_cap == f && ng == 0u) {
                    // (one cell alone always fits -- the host checks the parameters; whatever the arrays hold, the walk moves on)
                    const uint32_t nf = (uint32_t)__builtin_amdgcn_readlane((int)nn, f);
                    ng = min(nf, (uint32_t)VIEWS_BLOCK);
                    cs = cs_t;
                    cm = __builtin_amdgcn_readlane(M, f);
                    rm = __builtin_amdgcn_readlane(R, f);
                    const uint32_t rest = nf - ng;
                    close();
                    left = rest;
                    at = rest ? f : f + 1;
                    continue;
                }
                if (e_cap > f) {
                    // the last cell with groups before e_cap
                    const unsigned long long before = nz & ((1ull << e_cap) - 1ull);
                    const int p = 63 - __builtin_clzll(before);
                    ng = (uint32_t)__builtin_amdgcn_readlane((int)C, p);
                    cm = __builtin_amdgcn_readlane(M, p);
                    rm = __builtin_amdgcn_readlane(R, p);
                    cs = cs_t;
                }
                close();
                at = e_cap;
                left = (uint32_t)__builtin_amdgcn_readlane((int)nn, e_cap);
            } else if (e_cnt < 64) {
                // lane e_cnt fills the tile: it takes what is missing, the rest of the cell opens the next tile
                const uint32_t c_e = (uint32_t)__builtin_amdgcn_readlane((int)C, e_cnt);
                cm = __builtin_amdgcn_readlane(M, e_cnt);
                rm = __builtin_amdgcn_readlane(R, e_cnt);
                cs = cs_t;
                ng = (uint32_t)VIEWS_BLOCK;
                close();
                left = c_e - (uint32_t)VIEWS_BLOCK;
                at = left ? e_cnt : e_cnt + 1;
            } else {
                // everything left in the window joins the open tile
                ng = (uint32_t)__builtin_amdgcn_readlane((int)C, 63);
                cm = __builtin_amdgcn_readlane(M, 63);
                rm = __builtin_amdgcn_readlane(R, 63);
                cs = cs_t;
                break;
            }
        }
    }
    if (ng)
        close();
    if (!WRITE && lane == 0) {
        B.band_tiles[b] = tiles;
        B.band_groups[b] = placed;
        B.band_cost[b] = cost;
    }
}

__global__ __launch_bounds__(1024) void band_scan_kernel(BandParams B)
{
    // exclusive running sums over the bands: tiles, groups, and the cut's cost (for band_xcd_kernel)
    __shared__ uint32_t s_t[1024], s_g[1024];
    __shared__ unsigned long long s_c[1024];
    const int t = threadIdx.x;
    uint32_t run_t = 0u, run_g = 0u;
    unsigned long long run_c = 0ull;
    for (int base = 0; base < B.g.n_bands; base += 1024) {
        const int i = base + t;
        const uint32_t vt = i < B.g.n_bands ? B.band_tiles[i] : 0u, vg = i < B.g.n_bands ? B.band_groups[i] : 0u;
        const unsigned long long vc = i < B.g.n_bands ? B.band_cost[i] : 0ull;
        s_t[t] = vt; s_g[t] = vg; s_c[t] = vc;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint32_t at = t >= d ? s_t[t - d] : 0u, ag = t >= d ? s_g[t - d] : 0u;
            const unsigned long long ac = t >= d ? s_c[t - d] : 0ull;
            __syncthreads();
            s_t[t] += at; s_g[t] += ag; s_c[t] += ac;
            __syncthreads();
        }
        if (i < B.g.n_bands) {
            B.band_tiles[i] = run_t + s_t[t] - vt;
            B.band_groups[i] = run_g + s_g[t] - vg;
            B.band_cost[i] = run_c + s_c[t] - vc;
        }
        run_t += s_t[1023]; run_g += s_g[1023]; run_c += s_c[1023];
        __syncthreads();
    }
    if (t == 0) {
        B.info->n_tiles = run_t;
        B.info->n_groups = run_g;
        B.info->pad[0] = (uint32_t)run_c;
        B.info->pad[1] = (uint32_t)(run_c >> 32);
        if (B.host_words != nullptr) {
            // what the host waits for, straight into its page-locked block: no copy, no synchronisation of the stream (the plan
            // pass's gather counter was settled a kernel ago); the flag goes last, behind the acknowledged three
            __hip_atomic_store(B.host_words + 0, run_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(B.host_words + 1, run_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(B.host_words + 2, __hip_atomic_load(B.n_gather, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(B.host_words + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256) void band_scatter_kernel(BandParams B, uint32_t n_all)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_all)
        return;
    const uint32_t cell = B.gcell[g];
    if (cell == ~0u)
        return;
    const uint32_t pos = B.cell_off[cell] + atomicAdd(&B.cell_cur[cell], 1u);
    if (pos < (uint32_t)B.n_groups)
        B.sorted[pos] = g;
}

__global__ __launch_bounds__(VIEWS_BLOCK) void band_build_kernel(BandParams B)
{
    __shared__ uint32_t s_key[VIEWS_BLOCK];
    __shared__ int s_red[4][VIEWS_BLOCK / 64];
    const int t = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const BandTileRec rec = B.recs[tile];
    uint32_t cnt = rec.gcount < (uint32_t)VIEWS_BLOCK ? rec.gcount : (uint32_t)VIEWS_BLOCK;
    if (rec.gstart >= (uint32_t)B.n_groups)
        cnt = 0u;
    else if (rec.gstart + cnt > (uint32_t)B.n_groups)
        cnt = (uint32_t)B.n_groups - rec.gstart;
    s_key[t] = (uint32_t)t < cnt ? B.sorted[rec.gstart + t] : ~0u;
    __syncthreads();
    // bitonic sort, ascending: (pitch view, row, column) order, lanes without a group last
    for (int k = 2; k <= VIEWS_BLOCK; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            const int o = t ^ jj;
            if (o > t) {
                const uint32_t x = s_key[t], y = s_key[o];
                const bool up = (t & k) == 0;
                if ((x > y) == up) {
                    s_key[t] = y;
                    s_key[o] = x;
                }
            }
            __syncthreads();
        }
    const uint32_t g = s_key[t];
    const uint32_t gxn = (uint32_t)((B.ow + 3) >> 2), per_view = gxn * (uint32_t)B.oh;
    const bool has = g < per_view * (uint32_t)B.n_pitch;
    uint32_t pitch_i = 0u, y = 0u, x4 = 0u;
    int ix[4], iy[4];
    uint32_t fx[4], fy[4];
    bool live[4];
    int bx0 = INT32_MAX, bx1 = INT32_MIN, by0 = INT32_MAX, by1 = INT32_MIN;
    if (has) {
        pitch_i = g / per_view;
        const uint32_t rem = g - pitch_i * per_view;
        y = rem / gxn;
        x4 = rem - y * gxn;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        live[i] = false;
        ix[i] = iy[i] = 0;
        fx[i] = fy[i] = 0u;
        const int x = (int)(4u * x4) + i;
        if (has && x < B.ow) {
            const int2 c = B.coords[((size_t)pitch_i * B.oh + y) * B.ow + x];
            ix[i] = sat_short(c.x >> 5);
            iy[i] = sat_short(c.y >> 5);
            fx[i] = (uint32_t)c.x & 31u;
            fy[i] = (uint32_t)c.y & 31u;
            // all four taps inside the panorama (the plan pass has checked that for the tile): a pixel that is not, is dead
            live[i] = ix[i] >= 0 && iy[i] >= 0 && ix[i] + 1 < B.pw && iy[i] + 1 < B.ph;
            if (live[i]) {
                bx0 = min(bx0, ix[i]); bx1 = max(bx1, ix[i] + 1);
                by0 = min(by0, iy[i]); by1 = max(by1, iy[i] + 1);
            }
        }
    }
    bx0 = wave_reduce<false>(bx0); bx1 = wave_reduce<true>(bx1);
    by0 = wave_reduce<false>(by0); by1 = wave_reduce<true>(by1);
    if ((t & 63) == 0) {
        s_red[0][t >> 6] = bx0; s_red[1][t >> 6] = bx1; s_red[2][t >> 6] = by0; s_red[3][t >> 6] = by1;
    }
    __syncthreads();
    int c0 = INT32_MAX, cmax1 = INT32_MIN, r0 = INT32_MAX, rmax1 = INT32_MIN;
#pragma unroll
    for (int w = 0; w < VIEWS_BLOCK / 64; ++w) {
        c0 = min(c0, s_red[0][w]); cmax1 = max(cmax1, s_red[1][w]);
        r0 = min(r0, s_red[2][w]); rmax1 = max(rmax1, s_red[3][w]);
    }
    const bool any = cmax1 >= c0;
    uint32_t row_items = any ? (uint32_t)(((3 + cmax1 - c0) >> 2) + 1) : 1u;
    uint32_t rows = any ? (uint32_t)(rmax1 - r0 + 1) : 1u;
    // (the cut's conservative rectangle fits, so this one does; a garbage table must not make an oversized tile)
    const bool fits = row_items * rows <= (uint32_t)LDS_ITEMS_CAP && row_items < 65536u && r0 < 65536;
    if (!fits) {
        row_items = 1u;
        rows = 1u;
    }
    uint32_t* pxw = B.px + (size_t)tile * (VIEWS_BLOCK * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t word = 0u;
        if (live[i] && fits && any)
            word = (4u * row_items * (uint32_t)(iy[i] - r0) + (uint32_t)(ix[i] - c0)) | 1u << PXW_UP_BITS | fx[i] << 22 | fy[i] << 27;
        pxw[i * VIEWS_BLOCK + t] = word;
    }
    B.grp[(size_t)tile * VIEWS_BLOCK + t] =
        has ? (uint32_t)((size_t)pitch_i * B.view_bytes + (size_t)y * (size_t)B.out_row + 12u * (size_t)x4) : ~0u;
    if (t == 0) {
        PieceHdr h;
        h.mode_items = 3u | (any && fits ? row_items * rows : 0u) << 8;
        h.c0 = any ? c0 : 0;
        h.c1 = any ? cmax1 - 1 : -1;
        h.rows = (any ? (uint32_t)r0 : 0u) | row_items << 16;
        B.hdr[tile] = h;
    }
}

// The tiles (band order = source order) cut into eight runs of equal work, one per XCD: by the bands' costs (the cut's
// estimate: cost_base + the rectangle's items per tile), then tile by tile inside the band a boundary falls into.
__global__ __launch_bounds__(64) void band_xcd_kernel(BandParams B)
{
    const int lane = threadIdx.x;
    const uint32_t n = (uint32_t)B.n_tiles;
    const unsigned long long total = (unsigned long long)B.info->pad[0] | (unsigned long long)B.info->pad[1] << 32;
    if (lane >= 1 && lane < 8) {
        // band_cost[] holds the cost of the bands BEFORE each band (band_scan_kernel): binary search for the band the
        // boundary falls into, then a share of its tiles in proportion (a band's tiles cost about the same)
        const unsigned long long want = total * (unsigned long long)lane / 8ull;
        int lo = 0, hi = B.g.n_bands - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (B.band_cost[mid] <= want)
                lo = mid;
            else
                hi = mid - 1;
        }
        const unsigned long long before = B.band_cost[lo], after = lo + 1 < B.g.n_bands ? B.band_cost[lo + 1] : total;
        const uint32_t t0 = B.band_tiles[lo], t1 = lo + 1 < B.g.n_bands ? B.band_tiles[lo + 1] : n;
        uint32_t first = t0;
        if (after > before)
            first = t0 + (uint32_t)((unsigned long long)(t1 - t0) * (want - before) / (after - before));
        B.info->first[lane] = total ? first : n * (uint32_t)lane / 8u;
    }
    __syncthreads();
    if (lane == 0) {
        // no run longer than the view kernel's grid provides for
        uint32_t f[9];
        f[0] = 0u;
        f[8] = n;
        const uint32_t cap = (uint32_t)B.per_cap;
        for (uint32_t x = 1; x < 8; ++x) {
            uint32_t v = min(B.info->first[x], n);
            v = max(v, f[x - 1]);
            v = min(v, f[x - 1] + cap);
            const uint32_t need = (8u - x) * cap;  // tiles the XCDs from x on can still take
            if (n > need)
                v = max(v, n - need);
            f[x] = v;
        }
        for (uint32_t x = 0; x < 9; ++x)
            B.info->first[x] = f[x];
    }
    __syncthreads();
    // an XCD draws its run from the costlier end (towards a pole): the workgroups in flight when it runs out are its
    // cheapest -- compare the rectangles of the run's first and last tiles
    if (lane < 8) {
        const uint32_t a = B.info->first[lane], e = B.info->first[lane + 1];
        uint32_t rev = 0u;
        if (e > a + 8u) {
            unsigned long long head = 0ull, tail = 0ull;
            for (uint32_t i = 0; i < 4u; ++i) {
                const BandTileRec r = B.recs[a + i], q = B.recs[e - 1u - i];
                head += (uint32_t)((((3 + r.cmax1 - r.c0) >> 2) + 1) * (r.rmax1 - r.r0 + 1));
                tail += (uint32_t)((((3 + q.cmax1 - q.c0) >> 2) + 1) * (q.rmax1 - q.r0 + 1));
            }
            rev = tail > head ? 1u : 0u;
        }
        B.info->reversed[lane] = rev;
    }
}

hipError_t launch_band(const BandParams& B, int stage, hipStream_t st)
{
    if (stage == 0) {
        hipLaunchKernelGGL(band_cut_kernel<false>, dim3(B.g.n_bands), dim3(64), 0, st, B);
        hipLaunchKernelGGL(band_scan_kernel, dim3(1), dim3(1024), 0, st, B);
        return hipGetLastError();
    }
    if (B.n_tiles <= 0)
        return hipSuccess;
    const uint32_t n_all = (uint32_t)B.n_pitch * (uint32_t)B.oh * (uint32_t)((B.ow + 3) >> 2);
    hipLaunchKernelGGL(band_cut_kernel<true>, dim3(B.g.n_bands), dim3(64), 0, st, B);
    hipLaunchKernelGGL(band_scatter_kernel, dim3((n_all + 255u) / 256u), dim3(256), 0, st, B, n_all);
    hipLaunchKernelGGL(band_build_kernel, dim3(B.n_tiles), dim3(VIEWS_BLOCK), 0, st, B);
    hipLaunchKernelGGL(band_xcd_kernel, dim3(1), dim3(64), 0, st, B);
    return hipGetLastError();
}

// The plan pass: one workgroup per (tile, pitch view).  The workgroup that finishes LAST hands the number of gather tiles to
// the host (PlanParams::n_gather_host: page-locked memory the device writes through its own mapping): job_build_plan waits
// for the plan pass's event and reads a word, with the main kernel already queued behind the pass -- no copy in the
// stream, no waiting for that kernel, no kernel of its own for one word (a cold image's device side: 40 us of 190).
// Who is last: every workgroup adds one to its group's counter (workgroup & 63), the one that completes a group adds one
// to the groups' counter, the one that completes THAT has seen everybody: a thread's adds are issued one after the other,
// each behind the returned value of the one before (and behind the returned list index of its own gather tile: vmcnt(0)),
// and a returned value means the add has been performed where all of them are.  No fence: a release at agent scope writes
// an XCD's L2 back (the pass: 249 us instead of 52).  The last workgroup leaves every counter zero for the next pass.
template <bool CALLER_MAPS>
__global__ __launch_bounds__(VIEWS_BLOCK) void plan_kernel(PlanParams P)
{
    plan_tile<CALLER_MAPS>(P);
    if (P.n_gather_host == nullptr || threadIdx.x != 0)
        return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this workgroup's gather-list index has come back)
    const uint32_t total = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t g = b & (uint32_t)(PLAN_TICKET_GROUPS - 1);
    const uint32_t in_group = total / PLAN_TICKET_GROUPS + (g < total % PLAN_TICKET_GROUPS ? 1u : 0u);
    if (atomicAdd(P.ticket + 32 * (1 + g), 1u) != in_group - 1u)
        return;
    const uint32_t groups = total < (uint32_t)PLAN_TICKET_GROUPS ? total : (uint32_t)PLAN_TICKET_GROUPS;
    if (atomicAdd(P.ticket, 1u) != groups - 1u)
        return;
    const uint32_t n = atomicExch(P.n_gather, 0u);
    __hip_atomic_store(P.n_gather_host, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int k = 0; k <= PLAN_TICKET_GROUPS; ++k)
        __hip_atomic_store(P.ticket + 32 * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The quantised coordinates of EVERY pixel (plan_kernel with coords_all == 0 keeps only the gather tiles'): the same
// evaluation, the same rounding.
template <bool CALLER_MAPS>
__global__ __launch_bounds__(256) void coords_kernel(PlanParams P)
{
    const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y, pitch_i = blockIdx.z;
    if (px >= P.ow)
        return;
    const size_t k = ((size_t)pitch_i * P.oh + py) * P.ow + px;
    float U, V;
    if (CALLER_MAPS) {
        U = P.mapU[k];
        V = P.mapV[k];
    } else {
        const PitchConst pc = P.pitch[pitch_i];
        pitch_map_eval((float)px, (float)py, P.geom, pc.c, pc.s, U, V);
    }
    P.coords[k] = make_int2(cv_round_f32(U * 32.0f), cv_round_f32(V * 32.0f));
}

hipError_t launch_plan(const PlanParams& P, hipStream_t st)
{
    if (P.coords_only) {
        const dim3 grid((P.ow + 255) / 256, P.oh, P.n_pitch);
        if (P.mapU)
            hipLaunchKernelGGL(coords_kernel<true>, grid, dim3(256), 0, st, P);
        else
            hipLaunchKernelGGL(coords_kernel<false>, grid, dim3(256), 0, st, P);
        return hipGetLastError();
    }
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
#ifdef P2P_SHAPE_NO_FLOAT
    static const ShapeOps ops = {{P2P_SHAPE_NS::TILE_W, P2P_SHAPE_NS::TILE_H, P2P_SHAPE_NS::VIEWS_BLOCK, P2P_SHAPE_NS::VIEWS_PXT, P2P_SHAPE_NS::LDS_ITEMS_CAP},
                                 &P2P_SHAPE_NS::launch_plan, &P2P_SHAPE_NS::launch_remap_views, nullptr, &P2P_SHAPE_NS::launch_band,
                                 &P2P_SHAPE_NS::launch_pair_ctx};
#else
    static const ShapeOps ops = {{P2P_SHAPE_NS::TILE_W, P2P_SHAPE_NS::TILE_H, P2P_SHAPE_NS::VIEWS_BLOCK, P2P_SHAPE_NS::VIEWS_PXT, P2P_SHAPE_NS::LDS_ITEMS_CAP},
                                 &P2P_SHAPE_NS::launch_plan, &P2P_SHAPE_NS::launch_remap_views, &P2P_SHAPE_NS::launch_float_views, &P2P_SHAPE_NS::launch_band,
                                 &P2P_SHAPE_NS::launch_pair_ctx};
#endif
    return ops;
}

}  // namespace p2p
