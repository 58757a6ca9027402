// Stand-ins for the kernel launchers of p2p_device.h in the host-only sanitizer build (tests/test_sanitizers.py):
// they touch the buffers the host code reads back afterwards (piece headers, counters, yaw descriptors) with the
// sizes the real kernels use, so that ASan sees every host-side allocation being addressed, and do no pixel work.
#define P2P_HOST 1
#include "p2p_device.h"
#include <algorithm>
#include <string.h>
#include <vector>
extern "C" int p2p_stub_device_count = 1;
extern "C" int p2p_stub_drop_count = 0;  // 1: the plan pass "loses" the word it hands to the host (job_build_plan falls back to the headers)
#include <atomic>
static std::atomic<long> g_stub_live[2];
extern "C" long p2p_stub_live(int what) { return g_stub_live[what & 1].load(); }
extern "C" void p2p_stub_count(int what, long delta) { g_stub_live[what & 1].fetch_add(delta); }

namespace p2p {
constexpr int TILE_H = 16;  // rows of a tile in both shapes the library ships (p2p_device.h: P2P_TILE_ROWS)
hipError_t launch_yaw_tables(uint32_t* packed, float* rows, int pw, int n_yaw, const double* yaw_rad, hipStream_t)
{
    for (int y = 0; y < n_yaw; ++y)
        for (int c = 0; c < pw; ++c) {
            if (packed) packed[(size_t)y * pw + c] = 3u * (uint32_t)c;
            if (rows) rows[(size_t)y * pw + c] = (float)c + (float)(yaw_rad[y] * 0.0);
        }
    return hipSuccess;
}
hipError_t launch_zero_words(uint32_t* p, uint32_t n, hipStream_t)
{
    memset(p, 0, (size_t)n * sizeof(uint32_t));
    return hipSuccess;
}
// the main kernel's per-XCD lists: the host algorithm the device kernel (csrc/p2p_lists.hip) replaced -- counting sort by
// source band, running costs, eight cuts of equal work, every run from its costlier end
hipError_t launch_main_lists(const MainListParams& M, hipStream_t)
{
    auto band_of = [](const PieceHdr& h) { return (size_t)((((h.rows & 0xFFFFu) + (h.rows >> 16)) / 2u) >> 6); };
    std::vector<size_t> start(1026, 0);
    for (uint32_t s = 0; s < M.slots; ++s)
        if ((M.hdr[s].mode_items & 3u) == 1u)
            start[std::min<size_t>(band_of(M.hdr[s]), 1023) + 1]++;
    for (size_t b = 1; b < start.size(); ++b)
        start[b] += start[b - 1];
    const size_t n = start.back();
    for (uint32_t s = 0; s < M.slots; ++s)
        if ((M.hdr[s].mode_items & 3u) == 1u) {
            const size_t pos = start[std::min<size_t>(band_of(M.hdr[s]), 1023)]++;
            M.order[pos] = s;
            M.cost[pos] = M.cost_base + (M.hdr[s].mode_items >> 8);
        }
    std::vector<uint64_t> upto(n + 1, 0);
    for (size_t i = 0; i < n; ++i)
        upto[i + 1] = upto[i] + M.cost[i];
    size_t first[9];
    first[0] = 0;
    first[8] = n;
    for (int x = 1; x < 8; ++x) {
        size_t v = 0;
        while (v < n && upto[v] < upto[n] * (uint64_t)x / 8u)
            ++v;
        v = std::max(v, first[x - 1]);
        v = std::min(v, first[x - 1] + M.cap);
        if (n > (size_t)(8 - x) * M.cap)
            v = std::max(v, n - (size_t)(8 - x) * M.cap);
        first[x] = v;
    }
    for (int x = 0; x < 8; ++x) {
        const size_t a = first[x], b = first[x + 1], q = (b - a) / 4;
        const bool reversed = q > 0 && (upto[b] - upto[b - q]) > (upto[a + q] - upto[a]);
        for (size_t e = 0; e < M.cap; ++e)
            M.table[(size_t)x * M.cap + e] = e < b - a ? M.order[reversed ? b - 1 - e : a + e] : ~0u;
        M.count[x] = (uint32_t)(b - a);
    }
    return hipSuccess;
}
hipError_t launch_yaw_pack(uint32_t* packed, const float* rows, size_t n, hipStream_t)
{
    for (size_t k = 0; k < n; ++k) packed[k] = 3u * (uint32_t)rows[k];
    return hipSuccess;
}
hipError_t launch_yaw_desc(YawDesc* desc, uint32_t* f4tab, uint32_t* packed, int pw, int n_yaw, const double* yaw_rad, hipStream_t st)
{
    if (yaw_rad)  // (the descriptor kernel makes the packed table as it goes)
        (void)launch_yaw_tables(packed, nullptr, pw, n_yaw, yaw_rad, st);
    for (int y = 0; y < n_yaw; ++y) {
        desc[y] = YawDesc{(int)(packed[(size_t)y * pw] / 3u), y % 3 == 2 ? 1 : 0, 0, -1};
        memset(f4tab + (size_t)y * pw, 0, (size_t)pw * sizeof(uint32_t));
    }
    return hipSuccess;
}
hipError_t launch_rot_map(float* U, float* V, int ow, int oh, const MapGeom&, const float*, hipStream_t)
{
    memset(U, 0, (size_t)ow * oh * sizeof(float)); memset(V, 0, (size_t)ow * oh * sizeof(float));
    return hipSuccess;
}
hipError_t launch_pitch_map(float* U, float* V, int ow, int oh, const MapGeom&, float, float, hipStream_t)
{
    memset(U, 0, (size_t)ow * oh * sizeof(float)); memset(V, 0, (size_t)ow * oh * sizeof(float));
    return hipSuccess;
}
// the plan pass of one tile shape: every tile marked for gathers, the last word of every table touched
template <int TILE_W, int BLOCK, int CAP>
static hipError_t stub_plan(const PlanParams& P, hipStream_t)
{
    constexpr int PXT = TILE_W * TILE_H / BLOCK;
    const size_t tiles = (size_t)((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    if (P.coords_only) {  // ensure_full_coords: every pixel's coordinates into an existing plan
        memset(P.coords, 0, (size_t)P.n_pitch * P.oh * P.ow * sizeof(int2));
        return hipSuccess;
    }
    if (P.band.gcell) {  // band plan: no px / items tables; every group gets a cell (or ~0), the cells their counts
        const size_t gxn = (size_t)((P.ow + 3) / 4), n_groups = (size_t)P.n_pitch * P.oh * gxn;
        const size_t cells = (size_t)P.band.g.n_bands * P.band.g.ncx;
        uint32_t n_gather = 0;
        for (size_t s = 0; s < tiles * P.n_pitch; ++s) {
            const bool gathers = s % 5 == 0;
            P.hdr[s] = PieceHdr{gathers ? 2u : 3u, 0, 7, 0x00020001u};
            if (P.hdr_host)
                P.hdr_host[s] = P.hdr[s];
            if (gathers)
                P.gather_list[n_gather++] = (uint32_t)s;
        }
        for (size_t g = 0; g < n_groups; ++g) {
            const uint32_t cell = g % 7 == 0 ? ~0u : (uint32_t)(g % cells);
            P.band.gcell[g] = cell;
            if (cell != ~0u)
                P.band.cell_count[cell]++;
        }
        P.band.cell_cmin[cells - 1] = INT32_MAX; P.band.cell_cmax1[cells - 1] = 1; P.band.cell_rmax1[cells - 1] = 1;
        P.band.cell_cur[cells - 1] = 0; P.band.cell_off[cells - 1] = 0;
        memset(P.coords, 0, (size_t)P.n_pitch * P.oh * P.ow * sizeof(int2));
        P.n_gather[0] = n_gather;  // (band plans read the count back with a copy; the host word is the per-view plans')
        return hipSuccess;
    }
    // a mix of LDS-scheme and gather tiles with pseudo-random footprints (wild ones too: the host's work-list builders
    // must cope with whatever the headers say), so that xcd_main_lists / xcd_lists run under the sanitizers
    uint32_t n_gather = 0, h = 12345u + (uint32_t)(P.ow * 131 + P.oh);
    for (size_t s = 0; s < tiles * P.n_pitch; ++s) {
        h = h * 1664525u + 1013904223u;
        const bool gathers = (h >> 13) % 3u == 0u;
        const int c0 = (int)((h >> 8) % 70000u) - 2000, c1 = c0 + (int)((h >> 3) % 40000u) - 10;
        const uint32_t rows = (h % 5u == 0u) ? 0xFFFFFFFFu : ((h >> 4) & 0x0FFF0FFFu);
        P.hdr[s] = PieceHdr{(gathers ? 2u : 1u) | ((h >> 20) & 4u) | (gathers ? 0u : ((h >> 7) % (uint32_t)(CAP + 1)) << 8), c0, c1, rows};
        P.px[s * BLOCK * PXT + BLOCK * PXT - 1] = 0u;
        P.items[s * CAP + CAP - 1] = 0u;
        if (gathers)
            P.gather_list[n_gather++] = (uint32_t)s;
    }
    memset(P.coords, 0, (size_t)P.n_pitch * P.oh * P.ow * sizeof(int2));
    P.n_gather[0] = n_gather;
    if (P.n_gather_host && !p2p_stub_drop_count) {  // (the plan pass's last workgroup hands the count over itself and leaves the counters zero)
        *P.n_gather_host = n_gather;
        P.n_gather[0] = 0u;
    }
    return hipSuccess;
}
// the band passes: stage 0 counts (a few tiles, every banded group), stage 1 touches the last word of every table
template <int BLOCK>
static hipError_t stub_band(const BandParams& B, int stage, hipStream_t)
{
    if (stage == 0) {
        uint32_t groups = 0;
        const size_t n_all = (size_t)B.n_pitch * B.oh * ((B.ow + 3) / 4);
        for (size_t g = 0; g < n_all; ++g)
            groups += B.gcell[g] != ~0u;
        B.band_tiles[B.g.n_bands - 1] = 0; B.band_groups[B.g.n_bands - 1] = 0; B.band_cost[B.g.n_bands - 1] = 0;
        B.info->n_groups = groups;
        B.info->n_tiles = (groups + BLOCK - 1) / BLOCK;
        if (B.host_words && !p2p_stub_drop_count) {  // (band_scan_kernel hands the counts to the host itself)
            B.host_words[0] = B.info->n_tiles; B.host_words[1] = groups; B.host_words[2] = *B.n_gather; B.host_words[3] = 1u;
        }
        return hipSuccess;
    }
    const size_t nt = (size_t)B.n_tiles;
    B.recs[nt - 1] = BandTileRec{0, 0, 0, 1, 0, 1};
    B.sorted[B.n_groups > 0 ? B.n_groups - 1 : 0] = 0;
    for (size_t t = 0; t < nt; ++t)
        B.hdr[t] = PieceHdr{3u | 4u << 8, 0, 3, 2u << 16};
    B.px[nt * BLOCK * 4 - 1] = 0;
    B.grp[nt * BLOCK - 1] = ~0u;
    if ((int)((nt + 7) / 8) > B.per_cap)
        return hipErrorInvalidValue;
    for (int x = 0; x <= 8; ++x)
        B.info->first[x] = (uint32_t)(nt * x / 8);
    return hipSuccess;
}
// the pair-context table: its last record
static hipError_t stub_pair_ctx(const ViewsParams&, uint4* table, int slots, int chunks, int, hipStream_t)
{
    table[(size_t)slots * chunks * 64 - 1] = uint4{0u, 0u, 0u, 0u};
    return hipSuccess;
}
hipError_t launch_scramble(void*, size_t, uint32_t, hipStream_t) { return hipSuccess; }
hipError_t launch_compact_rows(void* dst, const uint8_t* src, size_t n_bytes, int row_bytes, int src_row, hipStream_t)
{
    for (size_t b = 0; b < n_bytes; ++b)
        ((uint8_t*)dst)[b] = src[(b / row_bytes) * src_row + b % row_bytes];
    return hipSuccess;
}
// a work list [8][stride] must hold every slot the headers give the kernel, once, and nothing else
static bool list_covers(const uint32_t* list, int stride, const PieceHdr* hdr, size_t slots, bool want_lds, bool want_gather)
{
    std::vector<int> seen(slots, 0);
    for (size_t i = 0; i < (size_t)8 * stride; ++i) {
        if (list[i] == ~0u)
            continue;
        if (list[i] >= slots || seen[list[i]]++)
            return false;
    }
    for (size_t s = 0; s < slots; ++s) {
        const uint32_t mode = hdr[s].mode_items & 3u;
        if (seen[s] != ((mode == 1u && want_lds) || (mode == 2u && want_gather) ? 1 : 0))
            return false;
    }
    return true;
}
template <int TILE_W>
static hipError_t stub_views(const ViewsParams& P, int which, hipStream_t)
{
    const size_t slots = (size_t)((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H) * P.n_pitch;
    if (which == 4) {  // the band kernel: its tables, and a grid that provides for every XCD's run
        if (P.band_tiles < 1 || !P.band_hdr || !P.band_px || !P.band_grp || !P.band_info || P.band_per < (P.band_tiles + 7) / 8)
            return hipErrorInvalidValue;
        if (P.band_hdr[P.band_tiles - 1].mode_items == 0u || P.band_info->first[8] != (uint32_t)P.band_tiles)
            return hipErrorInvalidValue;
    }
    if (which == 0 && P.main_list && (P.main_stride < 1 || P.main_group < 1 || P.main_chunks < 1 ||
                                      !list_covers(P.main_list, P.main_stride, P.hdr, slots, true, false)))
        return hipErrorInvalidValue;
    if (which == 0 && P.main_list) {
        // what the kernel's tail path indexes with: the count of every XCD's list, and a tail that fits
        for (int x = 0; x < 8; ++x) {
            int c = 0;
            while (c < P.main_stride && P.main_list[(size_t)x * P.main_stride + c] != ~0u)
                ++c;
            if (P.main_count[x] != c)
                return hipErrorInvalidValue;
        }
        if (P.main_span < 1 || P.main_tail < 0 || P.main_tail > P.main_stride ||
            (P.main_tail > 0 && (P.main_tail_parts < 2 || P.main_tail_parts > 4 || P.main_chunks != 1 || P.main_span != 1 || P.pf_lead != 0)))
            return hipErrorInvalidValue;
    }
    if (which == 3 && (P.n_list < 1 || !list_covers(P.gather_list, P.n_list, P.hdr, slots, P.gather_all != 0, true)))
        return hipErrorInvalidValue;
    if ((which == 2 || which == 1) && P.n_gather > 0)
        for (int i = 0; i < P.n_gather; ++i)
            if (which == 2 && P.gather_list[i] >= slots)
                return hipErrorInvalidValue;
    const size_t n = (size_t)P.n_panos * P.n_yaw * P.n_pitch * P.oh * P.ow * 3;
    P.out[0] = P.src[0];
    // the last byte of the last row's wrap pad: the upload's second 2-D copy must have filled it
    P.out[n - 1] = P.src[(size_t)(P.n_panos - 1) * P.pano_stride + (size_t)(P.ph - 1) * P.src_pitch +
                         3 * (P.pw + (P.pw < PANO_PAD ? P.pw : PANO_PAD)) - 1];
    return hipSuccess;
}
hipError_t launch_remap_maps(const RemapParams& P, int cn, int, hipStream_t)
{
    memset(P.dst, 0, (size_t)P.ow * P.oh * cn);
    return hipSuccess;
}
hipError_t launch_cubic_tab(short* tab, hipStream_t) { memset(tab, 0, 1024 * 16 * sizeof(short)); return hipSuccess; }
static hipError_t stub_float_views(const ViewsParams& P, bool, int, hipStream_t)
{
    P.out[(size_t)P.n_panos * P.n_yaw * P.n_pitch * P.oh * P.ow * 3 - 1] = 0;
    return hipSuccess;
}
const ShapeOps& shape_ops_w64()
{
    static const ShapeOps ops = {{64, TILE_H, 256, 4, 704}, &stub_plan<64, 256, 704>, &stub_views<64>, &stub_float_views, &stub_band<256>, &stub_pair_ctx};
    return ops;
}
const ShapeOps& shape_ops_w128()
{
    static const ShapeOps ops = {{128, TILE_H, 512, 4, 1408}, &stub_plan<128, 512, 1408>, &stub_views<128>, &stub_float_views, &stub_band<512>, &stub_pair_ctx};
    return ops;
}
const ShapeOps& shape_ops_w64b()
{
    static const ShapeOps ops = {{64, TILE_H, 256, 4, 960}, &stub_plan<64, 256, 960>, &stub_views<64>, nullptr, &stub_band<256>, &stub_pair_ctx};
    return ops;
}
}  // namespace p2p
