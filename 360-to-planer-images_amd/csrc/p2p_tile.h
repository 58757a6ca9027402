// p2p_tile.h -- what the tile-based view kernels share (p2p_views.hip: the reference's fixed-point arithmetic;
// p2p_float.hip: the opt-in float pixel path): tile geometry, the plan's item words, small types.
#ifndef P2P_TILE_H
#define P2P_TILE_H

#include "p2p_inline.h"
#include "p2p_audit.h"

namespace p2p {
namespace P2P_SHAPE_NS {

struct __attribute__((aligned(4))) Q16 { uint32_t d[4]; };

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
#ifndef P2P_STORE_AUX
#define P2P_STORE_AUX 2  // cache policy of the view stores: 2 = nt
#endif
// ... of the main kernel's (store_staged_pixels), per tile shape: 18 = nt sc1 for the 64-wide kernel, whose launches
// mostly stay inside the Infinity Cache (config 2 86.4 -> 84.7 us, config 5 743 -> 729 / 739 -> 735); the 128-wide
// kernel's streams lose 1-4 % with it, and so does the gather kernel (60.3 -> 69.1 us): both keep nt
// (profiles/r04_main_kernel_store_policy_all_configs.txt)
#ifndef P2P_MAIN_STORE_AUX_W64
#define P2P_MAIN_STORE_AUX_W64 18
#endif
#ifndef P2P_MAIN_STORE_AUX_W128
#define P2P_MAIN_STORE_AUX_W128 P2P_STORE_AUX
#endif
// Cache policy of the band kernel's stores: the default one (write-back).  A band tile's region of a view is cut where
// the SOURCE rectangle ends, at any 12-byte group: most 64-byte sectors of its rows are shared with the neighbouring
// tile, and under the streaming policies of the per-view tiles (nt, nt sc1: whole sectors there) every such sector goes
// out twice, half written -- config 2 235 us (nt sc1) / 215 (nt) / 131 (default); the L2 merges the two halves.
#ifndef P2P_BAND_STORE_AUX
#define P2P_BAND_STORE_AUX 0
#endif
#ifndef P2P_GATHER_LOAD_AUX
#define P2P_GATHER_LOAD_AUX 0  // ... of the gather kernel's taps (experiments)
#endif
#ifndef P2P_SRC_LOAD_AUX
#define P2P_SRC_LOAD_AUX 0  // cache policy of the main kernel's source pieces (experiments)
#endif

constexpr int TILE_LW = TILE_W == 128 ? 7 : (TILE_W == 64 ? 6 : (TILE_W == 32 ? 5 : 4));
static_assert((1 << TILE_LW) == TILE_W, "TILE_W must be 16, 32, 64 or 128");
constexpr int TILE_ROWSTEP = VIEWS_BLOCK / TILE_W;  // rows between a thread's pixels

// A tile's place in its view follows from its slot number alone (the workgroup's index, or a clamped work-list entry:
// never from the plan's tables); the plan's header adds
// how it is drawn (mode), the size of its footprint and its rot columns.
struct TileGeo {
    int x0, y0, pitch_i, mode, n_items;
    int blocky;  // gather kernel: its waves draw blocks of 16 x 4 pixels per instruction instead of rows of 64 (plan: see PieceHdr)
    int c0, c1;
    int col, row0;  // this thread's column and first row inside the tile
    uint32_t slot;  // pitch_i * tiles + tile: index of the tile's header, per-pixel words and item list
    int band_r0, band_row_items;  // band tiles: first rot row and items per row of the footprint rectangle
};

__device__ __forceinline__ TileGeo tile_geo(const ViewsParams& P, const PieceHdr& h, int pitch_i, int tile_id, int t)
{
    const int tiles_x = (P.ow + TILE_W - 1) / TILE_W;
    const int tiles = tiles_x * ((P.oh + TILE_H - 1) / TILE_H);
    TileGeo g;
    g.x0 = (tile_id % tiles_x) * TILE_W;
    g.y0 = (tile_id / tiles_x) * TILE_H;
    g.pitch_i = pitch_i;
    g.mode = (int)(h.mode_items & 3u);
    g.blocky = (int)((h.mode_items >> 2) & 1u);
    g.n_items = (int)(h.mode_items >> 8);
    g.c0 = h.c0;
    g.c1 = h.c1;
    g.col = t & (TILE_W - 1);
    g.row0 = t >> TILE_LW;
    g.slot = (uint32_t)pitch_i * (uint32_t)tiles + (uint32_t)tile_id;
    return g;
}

__device__ __forceinline__ int pano_of_pair(const ViewsParams& P, int pair)
{
    // pair -> panorama: multiply-high by ceil(2^32 / n_yaw) (exact for the job's sizes, host check); with one yaw
    // the constant would be 2^32, which does not fit, and the pair index is the panorama index anyway
    return P.n_yaw == 1 ? pair : (int)__umulhi((uint32_t)pair, P.n_yaw_magic);
}

// does the job draw view (yaw yi, pitch pitch_i)?  (yi < n_yaw, pitch_i < n_pitch: inside the mask by construction)
__device__ __forceinline__ bool view_wanted(const ViewsParams& P, int pitch_i, int yi)
{
    if (!P.view_mask || pitch_i < 0)  // (pitch_i < 0: a band tile's pair contexts -- its lanes test their own pitch views)
        return true;
    return ((P.view_mask[(size_t)pitch_i * P.mask_words + (yi >> 5)] >> (yi & 31)) & 1u) != 0u;
}

// pitch block of the grid -> pitch view, heaviest first; the table is the host's, its values are clamped all the same
__device__ __forceinline__ int pitch_of_block(const ViewsParams& P, int by)
{
    int p = P.pitch_order[by];
    P2P_AUD_LT(P.audit, AUD_MAIN_PITCH, p, P.n_pitch);
    return p < P.n_pitch ? p : P.n_pitch - 1;
}

// blockIdx.x -> tile, XCD-aware: the 8 XCDs each own a contiguous run of the tile raster, so neighbouring tiles
// (shared source halo and output lines) meet in one L2.  gridDim.x == 8 * ceil(tiles / 8); -1: no tile.
__device__ __forceinline__ int tile_of_block(const ViewsParams& P, int bx, int gx)
{
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const int chunk = gx >> 3;
    const int tile_id = (bx & 7) * chunk + (bx >> 3);
    return tile_id < tiles ? tile_id : -1;
}

// The main kernel's grid with table prefetch.  A workgroup starts with a chain of dependent loads (header -> per-pixel
// words, items, yaw descriptors -> first source pieces); when the plan tables of a job exceed the Infinity Cache
// (config 4: 113 MB per pitch view) their first touch in a launch comes from HBM, behind the launch's own write stream,
// and the workgroup -- four waves, 26 KB of LDS -- sits idle for tens of microseconds.  Loads complete in issue order
// per wave, so a drawing wave cannot fetch ahead for others without waiting for that fetch itself.  Hence: in every
// XCD's run of the tile raster, one workgroup in PF_GROUP + 1 draws nothing; it touches the tables of the PF_GROUP
// tiles that the same XCD starts pf_lead groups later (workgroups are dispatched to an XCD in index order), so those
// find them in L2.  gridDim.x == 8 * groups * (PF_GROUP + 1), groups = ceil(ceil(tiles / 8) / PF_GROUP).
constexpr int PF_GROUP = 32;
#ifdef P2P_ABLATE_HALF_PX
constexpr int PF_PX_LINES = VIEWS_BLOCK * VIEWS_PXT * 4 / 256;
#else
constexpr int PF_PX_LINES = VIEWS_BLOCK * VIEWS_PXT * 4 / 128;  // 128-byte lines of a tile's per-pixel words
#endif
#ifdef P2P_ABLATE_ITEMS_WINDOW
constexpr int PF_ITEM_LINES = 0;
#elif defined(P2P_PF_ITEM_LINES)
constexpr int PF_ITEM_LINES = P2P_PF_ITEM_LINES;
#else
constexpr int PF_ITEM_LINES = 12;                               // ... of its item list that are touched (384 items)
#endif
struct BlockRole {
    int tile_id;             // >= 0: draw this tile
    int pf_first, pf_count;  // pf_count > 0: touch the tables of tiles pf_first .. pf_first + pf_count - 1
};

__device__ __forceinline__ BlockRole main_block_role(const ViewsParams& P, int bx)
{
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const int per_xcd = (tiles + 7) >> 3;
    const int xcd = bx & 7, q = bx >> 3;
    BlockRole r;
    r.tile_id = -1;
    r.pf_first = r.pf_count = 0;
    int idx = q;
    if (P.pf_lead > 0) {
        const int g = q / (PF_GROUP + 1), s = q - g * (PF_GROUP + 1);
        idx = g * PF_GROUP + s - 1;
        if (s == 0) {
            const int first = (g + P.pf_lead) * PF_GROUP;
            int count = per_xcd - first;
            count = count < PF_GROUP ? count : PF_GROUP;
            if (xcd * per_xcd + first + count > tiles)
                count = tiles - (xcd * per_xcd + first);
            r.pf_first = xcd * per_xcd + first;
            r.pf_count = count > 0 ? count : 0;
            return r;
        }
    }
    const int tile_id = xcd * per_xcd + idx;
    if (idx < per_xcd && tile_id < tiles)
        r.tile_id = tile_id;
    return r;
}

__device__ __forceinline__ void decode_items(const uint32_t* __restrict__ itw, int t, int n_items, int src_pitch,
                                             uint32_t (&slot_off)[VIEWS_SLOTS], uint32_t (&slot_g)[VIEWS_SLOTS])
{
#pragma unroll
    for (int k = 0; k < VIEWS_SLOTS; ++k) {
        const int item = t + k * VIEWS_BLOCK;
        const uint32_t iw = itw[item < n_items && item < LDS_ITEMS_CAP ? item : 0];  // surplus lanes redo item 0 into LDS space nobody reads
        slot_g[k] = iw & 0xFFFFu;
        slot_off[k] = (iw >> 16) * (uint32_t)src_pitch + 12u * slot_g[k];  // rot row * src_pitch + 12 * g
    }
}

// Lanes 0..n-1 of a wave each hold one (panorama, yaw) pair's context and a class 0..NCLS-1; returns the lane every
// context has to move to so that the classes sit in ascending runs, and the run ends (pairs of classes 0..c) in cum[].
template <int NCLS>
__device__ __forceinline__ int sort_lanes_by_class(int k, bool valid, int cls, int (&cum)[NCLS])
{
    const unsigned long long below = (1ull << k) - 1ull;
    int r = k, base = 0;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        const unsigned long long m = __ballot(valid && cls == c);
        if (valid && cls == c)
            r = base + __popcll(m & below);
        base += __popcll(m);
        cum[c] = base;
    }
    return r;
}

}  // namespace P2P_SHAPE_NS
}  // namespace p2p
#endif  // P2P_TILE_H
