// p2p_tile.h -- what the tile-based view kernels share (p2p_views.hip: the reference's fixed-point arithmetic;
// p2p_float.hip: the opt-in float pixel path): piece geometry, the plan's item words, small types.
#ifndef P2P_TILE_H
#define P2P_TILE_H

#include "p2p_inline.h"

namespace p2p {

struct __attribute__((aligned(4))) Q16 { uint32_t d[4]; };


typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
#ifndef P2P_STORE_AUX
#define P2P_STORE_AUX 2  // cache policy of the view stores: 2 = nt
#endif


struct PieceGeo {
    int x0, y0, w, h, pitch_i, mode, n_items, lw;
    int col, row0, rstep;  // this thread's column and first row inside the piece; rows between its pixels
};

__device__ __forceinline__ PieceGeo piece_geo(const PieceHdr& h, int t)
{
    PieceGeo g;
    g.x0 = (int)(h.xy & 0xFFFFu);
    g.y0 = (int)(h.xy >> 16);
    g.w = (int)(h.geom & 0xFFu);
    g.h = (int)((h.geom >> 8) & 0xFFu);
    g.pitch_i = (int)(h.geom >> 16);
    g.mode = (int)(h.mode_items & 3u);
    g.n_items = (int)(h.mode_items >> 8);
    g.lw = __builtin_ctz((unsigned)g.w);
    g.col = t & (g.w - 1);
    g.row0 = t >> g.lw;
    g.rstep = VIEWS_BLOCK >> g.lw;
    return g;
}

__device__ __forceinline__ int pano_of_pair(const ViewsParams& P, int pair)
{
    // pair -> panorama: multiply-high by ceil(2^32 / n_yaw) (exact for the job's sizes, host check); with one yaw
    // the constant would be 2^32, which does not fit, and the pair index is the panorama index anyway
    return P.n_yaw == 1 ? pair : (int)__umulhi((uint32_t)pair, P.n_yaw_magic);
}


__device__ __forceinline__ void decode_items(const uint32_t* __restrict__ itw, int t, int n_items, int src_pitch,
                                             uint32_t (&slot_off)[VIEWS_SLOTS], uint32_t (&slot_g)[VIEWS_SLOTS])
{
#pragma unroll
    for (int k = 0; k < VIEWS_SLOTS; ++k) {
        const int item = t + k * VIEWS_BLOCK;
        const uint32_t iw = itw[item < n_items ? item : 0];  // surplus lanes redo item 0 into LDS space nobody reads
        slot_g[k] = iw & 0xFFFFu;
        slot_off[k] = (iw >> 16) * (uint32_t)src_pitch + 12u * slot_g[k];  // rot row * src_pitch + 12 * g
    }
}

}  // namespace p2p
#endif  // P2P_TILE_H
