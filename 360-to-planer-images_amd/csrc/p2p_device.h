// p2p_device.h -- structures and launchers shared by the device code (p2p_views / p2p_maps / p2p_remap / p2p_float .hip) and the
// host side of the C ABI (p2p_host.h: p2p_abi.cpp + p2p_host_*.cpp).  Not part of the public ABI (that is include/p2p_hip.h).
#ifndef P2P_DEVICE_H
#define P2P_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace p2p {

constexpr int PLAN_MAX_ROWS = 256;  // rot rows a tile's footprint may span (one plan thread per row)
// Every device panorama row is followed by a copy of the row's first PANO_PAD pixels: the source pixels of two
// neighbouring rot columns are then three contiguous pixels also where the yaw shift runs across the row's end
// (the gather kernel loads them with one 12-byte load).
constexpr int PANO_PAD = 8;
// gather kernel: a tile in which some output row of 64 pixels runs across this many source rows or more is drawn in
// blocks of 16 x 4 pixels per wave instruction (see PieceHdr::mode_items)
constexpr int GATHER_BLOCKY_FROM = 12;
constexpr int AUDIT_WORDS = 8;  // the audit build's violation record: hit, site, block x, y, z, thread, value, limit (p2p_audit.h)

// One tile of TILE_W x TILE_H output pixels of one pitch view, with everything that depends on the maps only worked
// out once by the plan pass.  mode 1 (LDS scheme): its footprint in the yaw-resampled panorama as a list of 4-pixel
// items (per-row spans) and, per pixel, the LDS offsets of its taps and its two 5-bit weights.  mode 2 (gathers): a
// footprint that does not fit the LDS buffers (strong minification, a pole inside the tile) or touches the
// panorama's border; the gather kernel draws it from the plan's quantised coordinates.  16 bytes, read with one
// scalar load; the tile's position and its pitch view come from its slot number (the workgroup's index, or a work-list
// entry clamped to the plan's slots), never from the tables.
struct PieceHdr {
    uint32_t mode_items;  // mode (1: LDS scheme, 2: gathers) | 4 if the gather kernel should draw 16 x 4 blocks (the rows of 64
                          // output pixels run ACROSS the source rows: next to a pole, every lane of a row in another line) | n_items << 8
    int32_t c0, c1;       // rot columns the taps of the tile's live pixels read: c0 .. c1 + 1 (c1 < c0: no live pixel).
                          // mode 1: LDS position 0 of every row is a column congruent to c0 mod 4
    uint32_t rows;        // (first rot row of the upper taps + 1) | (last + 1) << 16; 0: no live pixel.  Read by the host only
                          // (the order of the gather lists), never by a kernel
};
static_assert(sizeof(PieceHdr) == 16, "PieceHdr is read as one s_load_dwordx4");
// mode 3 (band plans only): the tile's pixels are drawn from source-band tiles (below), the header carries nothing else.

// ---- source-band tiles.  The dual of the tiles above: a tile is a RECTANGLE of the yaw-resampled panorama (a band of
// rot rows x a run of columns) and the set of 4-pixel output groups -- of ANY pitch view -- whose taps fall into it.
// Views whose footprints overlap (config 2's 60 / 90 / 120 degree views by half) share stage 1, a strongly minifying
// view set (the reference CLI's defaults) gets LDS tiles at all, and a lane draws the 4 adjacent pixels it stores: no
// staging.  Built on the device by the band passes (p2p_plan.hip): groups -> cells of the source (counting sort) -> a
// greedy cut of every band's cell row into tiles of <= VIEWS_BLOCK groups and <= LDS_ITEMS_CAP items -> per tile the
// groups in (pitch view, row, column) order, one per lane.
// A band tile's header is a PieceHdr: mode_items = 3 | n_items << 8 (n_items = rows * row_items, a rectangle: no item
// list), c0 / c1 as above, rows = first rot row | items per row << 16.  Per pixel word: as above with the distance
// field reduced to a live flag (the lower tap is one LDS row further: 4 * row_items dwords).  Per group one word: byte
// offset of its 12 bytes inside the n_pitch views of a (panorama, yaw) pair, ~0 = lane without a group.
struct BandGeom {
    int bh, cw;        // cell of the source: bh rot rows x cw columns (a group belongs to the cell of its upper-left tap)
    int ncx, n_bands;  // cells per band, bands
    int maxw, maxh;    // a group whose taps span more columns / rows than this sends its tile to the gather kernel
};
struct BandTileRec {   // what the cut knows about a tile (24 bytes)
    uint32_t gstart, gcount;  // its groups: a run of the cell-sorted group list
    int32_t c0, cmax1;        // columns its taps may read: c0 .. cmax1 (conservative: the cells' extents)
    int32_t r0, rmax1;        // rows r0 .. rmax1
};
struct BandInfo {      // device-side summary, read back once per geometry
    uint32_t n_tiles, n_groups;
    uint32_t first[9];     // XCD x draws tiles first[x] .. first[x + 1] - 1 (equal work, at most per_cap each)
    uint32_t reversed[8];  // ... from the far end (its costlier end first)
    uint32_t pad[5];
};
struct BandParams {
    int pw, ph, ow, oh, n_pitch;
    // band_scan_kernel hands the counts the host sizes the band tables by to page-locked memory itself (nullptr: the host
    // copies BandInfo back): [0] tiles, [1] groups, [2] the plan pass's gather tiles (*n_gather), [3] 1 once the three are there
    uint32_t* host_words;
    const uint32_t* n_gather;
    BandGeom g;
    const int2* coords;       // the plan pass's quantised coordinates
    uint32_t* cell_count;     // [n_bands * ncx] groups per cell
    int* cell_cmin;           //   ... INT32_MAX - their leftmost tap column (0: no group)
    int* cell_cmax1;          //   ... rightmost tap column
    int* cell_rmax1;          //   ... lowest tap row
    uint32_t* cell_off;       //   ... first position in the sorted list
    uint32_t* cell_cur;       //   ... scatter cursor
    uint32_t* gcell;          // [n_pitch * oh * ceil(ow / 4)] cell of every group, ~0: its tile gathers
    uint32_t* band_tiles;     // [n_bands] tiles of the band, then (scan) its first tile
    uint32_t* band_groups;    // [n_bands] groups of the band, then its first position in the sorted list
    unsigned long long* band_cost;  // [n_bands] the cut's cost estimate of the band's tiles (band_xcd_kernel)
    BandInfo* info;
    BandTileRec* recs;        // [n_tiles]
    uint32_t* sorted;         // [n_groups] group indices, by cell
    PieceHdr* hdr;            // [n_tiles] band tile headers
    uint32_t* px;             // [n_tiles][VIEWS_BLOCK * 4]
    uint32_t* grp;            // [n_tiles][VIEWS_BLOCK]
    int n_tiles, n_groups;    // host copies (after the read-back)
    int out_row;
    size_t view_bytes;
    int per_cap;              // tiles per XCD the view kernel's grid provides for
    uint32_t cost_base;
};
// per-pixel word: tap_up (dwords into an LDS buffer, PXW_UP_BITS) | (tap_lo - tap_up) << 12 (PXW_DL_BITS, 0 = the
// pixel has no footprint: NaN coordinate) | fx << 22 | fy << 27
// item word: rot row << 16 | 4-pixel group relative to the group of column c0

// how the yaw map of one yaw angle acts on columns (see yaw_desc_kernel)
struct YawDesc {
    int s;        // rot column c reads source column (c + s) mod pw
    int mode;     // 0: one two-tap weight f for every column; 1: per-column weights (f4tab); 2: not a shift
    int f;        // mode 0: the weight, 0..32
    int c_clamp;  // mode 0: the one column whose weight differs (clipped to pw-1), or -1
};

// scalars of precompute_pitch_mapping (P:114-175) that NumPy evaluates once, on the host, in
// float64 and then uses as float32
struct MapGeom {
    float half_w, half_h;  // W / 2.0, H / 2.0          (P:129-130)
    float focal;           // 0.5 * W / tan(FOV / 2)    (P:119, cast at P:131)
    float pw_f, ph_f;      // panorama size as float32  (P:167-173)
};

struct PitchConst {
    float c, s;  // cos / sin of the pitch angle, float64 -> float32 (R_pitch dtype, P:142-149)
};

struct ViewsParams {
    const uint8_t* src;      // [n_panos] panoramas, each ph rows of src_pitch bytes (BGR interleaved, PANO_PAD wrap pixels)
    size_t pano_stride;      // bytes between panoramas (< 2^32: one buffer descriptor spans a panorama)
    int src_pitch;           // bytes between rows (>= 3 * (pw + PANO_PAD))
    int pw, ph;
    const uint32_t* ytab;    // [n_yaw][pw] packed yaw-table entries: 3*ix | fx << 20
    const YawDesc* ydesc;    // [n_yaw]
    const uint32_t* f4tab;   // [n_yaw][pw] weights of columns c..c+3 (mod pw), one byte each
    int n_yaw, n_pitch, n_panos;
    int pairs_per_block;     // (panorama, yaw) pairs looped over by one workgroup
    const uint32_t* main_list;  // main kernel: [8][main_stride] the LDS-scheme tiles dealt to the XCDs in source order (p2p_host_plan.cpp:
    int main_stride;            // xcd_main_lists), ~0 = none; nullptr: the grid's own (tile, chunk, pitch view) order
    int main_group, main_chunks;  // list entries an XCD draws for one chunk of pairs before it turns to the next chunk; chunks of pairs
                                  // (main_chunks counts workgroups per tile: chunks of pairs / main_span, rounded up)
    int main_span;           // main kernel: chunks of pairs ONE workgroup draws, one after the other (>= 1)
    int main_tail;           // list order, one chunk of pairs: the last main_tail entries of every XCD's list are drawn by
    int main_tail_parts;     // main_tail_parts workgroups each, a part of the pairs each (0 = off)
    const uint32_t* main_count;  // [8] entries of each XCD's list (written with the list by main_lists_kernel: the host knows only the stride)
    int pf_lead;             // main kernel: > 0 = every (PF_GROUP + 1)-th workgroup of an XCD's run draws nothing and touches the plan
                             // tables of the PF_GROUP tiles that start pf_lead groups later (p2p_tile.h: main_block_role)
    int chunk_outer;         // tile grids: 0 = (tile, chunk, pitch view), 1 = (tile, pitch view, chunk) -- see pair_chunk
    uint32_t n_yaw_magic;    // ceil(2^32 / n_yaw): pair / n_yaw == umulhi(pair, magic) while pair * n_yaw < 2^32
    const PitchConst* pitch; // [n_pitch]
    MapGeom geom;
    int ow, oh;
    uint8_t* out;            // [n_panos][n_yaw][n_pitch][oh] rows of out_row bytes ([ow][3] pixels, then padding)
    int out_row;             // bytes per view row on the device: 3 * ow rounded up to whole dwords x 3 (a multiple of 12),
                             // so that every row starts dword-aligned whatever the width; the host copies rows out
    size_t view_bytes;       // oh * out_row
    int border;              // stage-2 border mode (0 = BORDER_CONSTANT 0, the reference's current tool)
    const uint16_t* pitch_order;  // [n_pitch] blockIdx.y -> pitch index, heaviest view first
    const int2* coords;      // [n_pitch][oh][ow] quantised pitch-stage coordinates (sx, sy), written by the plan pass
    const PieceHdr* hdr;     // [n_pitch][tiles]
    const uint32_t* px;      // [n_pitch][tiles][256 * VIEWS_PXT]
    const uint32_t* items;   // [n_pitch][tiles][LDS_ITEMS_CAP]
    const uint32_t* px2;     // float pixel path: 16-bit coordinate fractions per pixel
    const double* yaw_rad;   // float pixel path: [n_yaw]
    const uint32_t* odd_pairs;    // the (panorama, yaw) pairs (pano * n_yaw + yaw) whose yaw is not a plain shift with one
    int n_odd_pairs;              // weight: the rest kernel's and the table kernel's pair list (see use_pair_list)
    int use_pair_list;            // 1: those two kernels draw only the listed pairs; 0: every pair of the job
    int rest_ppb;
    const uint32_t* gather_list;  // the plan's mode-2 tiles (pitch * tiles + tile), in no particular order, and how many
    int n_gather;
    int gather_ppb;          // (panorama, yaw) pairs per workgroup of the gather / table kernels
    int n_list;              // gather kernel: gather_list is [8][n_list], one work list per XCD, ~0 = no tile (p2p_host_plan.cpp: xcd_lists)
    int gather_all;          // 1: the gather kernel draws EVERY tile (few of the job's tiles fit the LDS scheme: one launch less)
    // sparse view sets (p2p_job_set_view_mask; the view-sharded multi-GPU path: a rank's 4-5 views of one image in ONE
    // launch): bit (yaw & 31) of word [pitch][yaw >> 5] = the view is drawn; nullptr: all of them.  A view that is not
    // wanted is neither computed (main / gather kernels: a pair class of its own, skipped) nor stored.
    const uint32_t* view_mask;
    int mask_words;          // words per pitch: ceil(n_yaw / 32)
    float centre;            // float pixel path only: 0 = the reference's sampling convention, 0.5 = pixel centres
    const uint4* pair_ctx;   // [slots][pair_ctx_chunks][64] the (panorama, yaw) pair contexts of every tile and chunk of pairs
    int pair_ctx_chunks;     // (p2p_views.hip: pair_ctx_kernel), or nullptr: every workgroup works its own out
    uint32_t* audit;         // -DP2P_AUDIT builds: the context's violation record (see p2p_audit.h); else nullptr
    // source-band tiles (band plans): the band kernel draws them for every plain-shift yaw instead of the main kernel
    const PieceHdr* band_hdr;
    const uint32_t* band_px;
    const uint32_t* band_grp;
    const BandInfo* band_info;
    int band_tiles;          // tiles (host copy); 0: no band plan
    int band_per;            // list entries per XCD the grid provides for (>= every first[x + 1] - first[x])
    int band_tail;           // like main_tail: the last band_tail tiles of every XCD's run are drawn by main_tail_parts workgroups
    const uint32_t* merge_gather_list;  // != nullptr: the plan's gather tiles ([8][merge_gather_n] per-XCD lists, as gather_list / n_list) are
    int merge_gather_n;                 // drawn by the first 8 * merge_gather_n * (chunks of gather_ppb pairs) workgroups of the band kernel's
                                        // launch, or of the main kernel's in list order, not by a launch of their own
};

// plan_kernel's finish counters: [0] groups that have finished, [32 * (1 + g)] workgroups of group g (workgroup & 63) that
// have -- 128 bytes apart: 6120 adds to ONE word are 37 us of a 52 us pass, 96 to each of 64 words are not seen
constexpr int PLAN_TICKET_GROUPS = 64, PLAN_TICKET_WORDS = 32 * (1 + PLAN_TICKET_GROUPS);

struct PlanParams {
    int pw, ph, ow, oh, n_pitch, border;
    MapGeom geom;
    const PitchConst* pitch;
    const float* mapU;       // caller maps [n_pitch][oh][ow] or nullptr (then pitch_map_eval)
    const float* mapV;
    int2* coords;
    int coords_all;          // 1: the coordinates of every pixel are written; 0: only those of the tiles that gather
    int ty0, ty1;            // the job draws the tile rows ty0 .. ty1 - 1 of every view (p2p_job_set_rows); the others get mode 0: nobody's
    int coords_only;         // 1: launch_plan runs coords_kernel instead (every pixel's coordinates into an existing plan)
    PieceHdr* hdr;
    uint32_t* px;
    uint32_t* items;
    int float_path;          // plan for the float pixel path: unclipped azimuth, 16-bit fractions in px2, spans one
    float centre;            // column wider (the yaw's fractional shift may carry); centre: 0 or 0.5 (pixel centres)
    uint32_t* px2;           // float path: [n_pitch][tiles][256 * VIEWS_PXT] frac(U) | frac(V) << 16, 1/65536 units
    int blocky_from;         // a tile with an output row of 64 pixels across this many source rows is drawn in 16 x 4 blocks (GATHER_BLOCKY_FROM)
    uint32_t* n_gather;      // [0] tiles marked for gathers
    // Per-view plans: the context's finish counters (PLAN_TICKET_WORDS words, zero between plan passes) and the page-locked
    // host word the pass's LAST workgroup writes the gather count to (plan_kernel); nullptr: nobody reads it there
    uint32_t* ticket;
    uint32_t* n_gather_host;
    PieceHdr* hdr_host;      // band plans: every tile's header a second time, in page-locked host memory (nullptr: not wanted)
    uint32_t* gather_list;   // [n_pitch * tiles] the tiles marked for gathers (pitch * tiles + tile), in no particular order
    // band plan (band.gcell != nullptr): a tile all of whose groups can go into source-band tiles gets mode 3 and no
    // tables (px / items are nullptr); its groups are counted into the cells of the source.  The other tiles gather.
    BandParams band;
};

struct RemapParams {
    const uint8_t* src;
    int sw, sh, src_pitch;
    const float* U;
    const float* V;
    uint8_t* dst;
    int ow, oh;
    int border;
    uint8_t cval[4];
    const short* ctab;  // INTER_CUBIC only: [32 * 32][4][4] fixed-point weights (cubic_tab_kernel)
};

// the main kernel's per-XCD work lists, built on the device from the plan's headers (p2p_lists.hip)
struct MainListParams {
    const PieceHdr* hdr;   // [slots] = [n_pitch][tiles]
    uint32_t slots;
    uint32_t cost_base;    // a tile costs about cost_base + its footprint's items
    uint32_t cap;          // entries per XCD the table (and the main kernel's grid) provides for; 8 * cap >= slots
    uint32_t* order;       // scratch [slots]: the mode-1 tiles by (source band, view, raster)
    uint32_t* cost;        // scratch [slots]: their costs in that order
    uint32_t* table;       // out [8][cap]: XCD x's entries, then ~0
    uint32_t* count;       // out [8]
};
hipError_t launch_main_lists(const MainListParams& M, hipStream_t st);
hipError_t launch_zero_words(uint32_t* p, uint32_t n, hipStream_t st);  // (instead of hipMemsetAsync in front of a kernel)

hipError_t launch_yaw_tables(uint32_t* packed, float* rows, int pw, int n_yaw, const double* yaw_rad,
                             hipStream_t st);
hipError_t launch_yaw_pack(uint32_t* packed, const float* rows, size_t n, hipStream_t st);
hipError_t launch_yaw_desc(YawDesc* desc, uint32_t* f4tab, uint32_t* packed, int pw, int n_yaw, const double* yaw_rad,
                           hipStream_t st);
hipError_t launch_rot_map(float* U, float* V, int ow, int oh, const MapGeom& g, const float* R9, hipStream_t st);
hipError_t launch_pitch_map(float* U, float* V, int ow, int oh, const MapGeom& g, float c, float s,
                            hipStream_t st);
hipError_t launch_scramble(void* p, size_t bytes, uint32_t seed, hipStream_t st);  // robustness self-test only
// padded device view rows (src_row bytes apart, row_bytes used) -> contiguous bytes, dst sized to whole dwords
hipError_t launch_compact_rows(void* dst, const uint8_t* src, size_t n_bytes, int row_bytes, int src_row, hipStream_t st);
hipError_t launch_remap_maps(const RemapParams& P, int cn, int interpolation, hipStream_t st);
hipError_t launch_cubic_tab(short* tab, hipStream_t st);

// ---- tile shapes.  The plan pass and the view kernels (p2p_plan / p2p_views / p2p_float .hip) are compiled once per
// tile shape, each in its own namespace:
//   w64   64 x 16 output pixels per workgroup of 256 threads, LDS buffers of 704 items (six workgroups per CU): the
//         shape of jobs whose views mostly stay in the Infinity Cache (config 2: 85 us against 95);
//   w128  128 x 16 pixels per workgroup of 512 threads, 1408 items: a wave's store covers whole 128-byte lines (2 rows x
//         384 bytes instead of 4 x 192), which is what counts when tens of gigabytes of views stream to HBM
//         (config 4: 6.5 ms against 7.2; config 2 loses 12 %, the CLI's default set 25 %).
// The host picks a shape per job (p2p_host_plan.cpp: choose_shape) and calls through ShapeOps.
struct TileShape {
    int tile_w, tile_h, block, pxt, cap;  // TILE_W, TILE_H, VIEWS_BLOCK, VIEWS_PXT, LDS_ITEMS_CAP of the shape
};
struct ShapeOps {
    TileShape shape;
    hipError_t (*plan)(const PlanParams& P, hipStream_t st);
    hipError_t (*views)(const ViewsParams& P, int which, hipStream_t st);
    hipError_t (*float_views)(const ViewsParams& P, bool half, int which, hipStream_t st);
    // the band passes after the plan pass: 0 = cut count + scan (then the host reads BandInfo back and allocates),
    // 1 = cut, scatter, tile build, XCD runs
    hipError_t (*band)(const BandParams& B, int stage, hipStream_t st);
    // the job's pair-context table for the main kernel's tiles (band = 0) or the band tiles (1)
    hipError_t (*pair_ctx)(const ViewsParams& P, uint4* table, int slots, int chunks, int band, hipStream_t st);
};
const ShapeOps& shape_ops_w64();
const ShapeOps& shape_ops_w128();
const ShapeOps& shape_ops_w64b();  // the band shape: 64 x 16 tiles, LDS buffers of 1000 items (p2p_views_band.hip)

#ifndef P2P_HOST  // device translation units: the constants of THEIR shape
#ifndef P2P_SHAPE_NS
#define P2P_SHAPE_NS w64
#endif
namespace P2P_SHAPE_NS {
#ifndef P2P_TILE_W
#define P2P_TILE_W 64
#endif
#ifndef P2P_WAVES
#define P2P_WAVES (P2P_TILE_W == 64 ? 7 : 6)  // 64 x 16: 7 workgroups x 22.5 KB of LDS (the staging dwords live inside the tile buffers)
#endif
#ifndef P2P_TILE_ROWS
#define P2P_TILE_ROWS 16
#endif
constexpr int TILE_W = P2P_TILE_W;  // output tile of one workgroup
constexpr int TILE_H = P2P_TILE_ROWS;
#ifndef P2P_BLOCK
#define P2P_BLOCK 256
#endif
constexpr int VIEWS_BLOCK = P2P_BLOCK;  // threads of a tile's workgroup
constexpr int VIEWS_PXT = TILE_W * TILE_H / VIEWS_BLOCK;  // output pixels per thread (rows ROWSTEP apart)
#ifndef P2P_SLOTS
#define P2P_SLOTS 3
#endif
#ifndef P2P_CAP
#define P2P_CAP 704  // 6 workgroups (waves per SIMD) fit the 160 KB of LDS; the largest tile footprint of config 2 is 678
#endif
constexpr int VIEWS_SLOTS = P2P_SLOTS;  // 16-byte footprint items one thread produces per (panorama, yaw) pair, at most
constexpr int LDS_ITEMS_CAP = P2P_CAP;  // items (4 rot pixels each) per LDS buffer
// (a wave produces a slot's 64 items or none of them -- stage 1 does not test the item index per lane: a buffer that ended
// inside a wave's 64 would have the wave's surplus lanes write into the other buffer, which slower waves still read)
static_assert(LDS_ITEMS_CAP % 64 == 0, "whole waves of items");
static_assert(LDS_ITEMS_CAP <= VIEWS_SLOTS * VIEWS_BLOCK && LDS_ITEMS_CAP > (VIEWS_SLOTS - 1) * VIEWS_BLOCK, "slots vs cap");
constexpr int PXW_UP_BITS = 4 * LDS_ITEMS_CAP <= 4096 ? 12 : 13;  // per-pixel word: bits of the upper tap's LDS offset (dwords)
constexpr int PXW_DL_BITS = 22 - PXW_UP_BITS;                     //                 bits of (lower tap - upper tap)
static_assert(4 * LDS_ITEMS_CAP <= (1 << PXW_UP_BITS), "tap offsets must fit the per-pixel word");
constexpr int VIEWS_WAVES_PER_SIMD = P2P_WAVES;  // __launch_bounds__ of the main view kernel: 80 VGPRs, and 6 x 26.5 KB of LDS
hipError_t launch_plan(const PlanParams& P, hipStream_t st);
hipError_t launch_remap_views(const ViewsParams& P, int which, hipStream_t st);
hipError_t launch_float_views(const ViewsParams& P, bool half, int which, hipStream_t st);
hipError_t launch_band(const BandParams& B, int stage, hipStream_t st);
hipError_t launch_pair_ctx(const ViewsParams& P, uint4* table, int slots, int chunks, int band, hipStream_t st);
}  // namespace P2P_SHAPE_NS
using namespace P2P_SHAPE_NS;
#define P2P_SHAPE_OPS_NAME2(ns) shape_ops_##ns
#define P2P_SHAPE_OPS_NAME1(ns) P2P_SHAPE_OPS_NAME2(ns)
#define P2P_SHAPE_OPS_NAME P2P_SHAPE_OPS_NAME1(P2P_SHAPE_NS)
#endif

}  // namespace p2p

#endif
