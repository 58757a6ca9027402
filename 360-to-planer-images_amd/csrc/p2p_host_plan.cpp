// p2p_host_plan.cpp -- how a job is drawn -- tile shape, pairs per workgroup, list order, band tiles or per-view tiles -- and the host side
// of the plan pass (p2p_plan.hip): tables, read-backs, the per-XCD work lists.
// Part of the host side of libp2p_hip.so (see p2p_host.h for the units); C ABI: include/p2p_hip.h via p2p_abi.cpp.
#include "p2p_host.h"

namespace p2p_host {

// Z-order of the centre of a tile's footprint in the SOURCE panorama, in cells of 64 columns x 32 rows (xcd_lists).
uint64_t source_order_key(const p2p::PieceHdr& h)
{
    if (h.c1 < h.c0 || h.rows == 0u)
        return ~0ull;  // no live pixel: reads nothing
    const uint32_t cx = (uint32_t)std::max(0, (h.c0 + h.c1) / 2) >> 6;
    const uint32_t cy = (((h.rows & 0xFFFFu) + (h.rows >> 16)) / 2u) >> 5;
    uint64_t key = 0;
    for (int b = 0; b < 16; ++b)
        key |= (uint64_t)((cx >> b) & 1u) << (2 * b) | (uint64_t)((cy >> b) & 1u) << (2 * b + 1);
    return key;
}

// The main kernel's per-XCD lists (p2p_lists.hip: main_lists_kernel) for a plan whose headers are on the device: one
// block [table 8 x cap | count 8 | order, cost: the kernel's scratch], one launch on `st`, nothing read back.  The table's
// stride is all the host knows: a quarter more than an equal share of ALL the plan's tiles (the runs are of equal work,
// not of equal length; the kernel keeps every run within it).
int plan_enqueue_main_lists(Plan& Pl, size_t slots, int tile_w, hipStream_t st)
{
    const size_t cap = (slots + 7) / 8 + (slots + 31) / 32 + 16;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_table = up(8 * cap * sizeof(uint32_t)), b_count = 256, b_scratch = up(slots * sizeof(uint32_t));
    unsigned char* blk = nullptr;
    HIP_TRY(dev_alloc((void**)&blk, b_table + b_count + 2 * b_scratch));
    p2p::MainListParams M{};
    M.hdr = Pl.d_hdr;
    M.slots = (uint32_t)slots;
    // equal WORK per XCD, not equal counts: a tile costs about 600 + its footprint's items (stage 2 and the way out,
    // plus stage 1 per item), and the footprints grow towards the poles -- with equal counts the two XCDs that hold the
    // polar bands finish last (config 3's share: 7.6 ms against 6.5 in grid order)
    // (the constant, swept: config 2, 64-wide tiles, is flat from 200 to 1400 -- 84.6 ... 85.1 us, 86.9 at 0, 86.0 at 3000;
    // config 4, 128-wide tiles of twice the pixels, has a sharp optimum: 450 / 525 / 600 / 675 / 750 / 850 / 1000 give
    // 6.32 / 6.26 / 6.22 / 6.17 / 6.25 / 6.32 / 6.45 ms)
    M.cost_base = tile_w == 128 ? 675u : 600u;
    M.cap = (uint32_t)cap;
    M.table = (uint32_t*)blk;
    M.count = (uint32_t*)(blk + b_table);
    M.order = (uint32_t*)(blk + b_table + b_count);
    M.cost = (uint32_t*)(blk + b_table + b_count + b_scratch);
    const hipError_t e = p2p::launch_main_lists(M, st);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        (void)dev_free(blk);
        return fail(P2P_ERR_HIP, "main lists: %s", hipGetErrorString(e));
    }
    Pl.d_main_list = M.table;
    Pl.d_main_count = M.count;
    Pl.main_stride = (int)cap;
    Pl.bytes += b_table + b_count + 2 * b_scratch;
    return P2P_OK;
}

// The gather kernel's tiles, dealt to the 8 XCDs (workgroup b runs on XCD b & 7 and takes entry b >> 3 of that XCD's
// list).  Views of different pitch read overlapping parts of the panorama -- the reference CLI's defaults draw five
// pitch views per yaw, each covering a quarter of it -- and every XCD has its own L2: a source line that tiles on
// several XCDs want crosses the fabric several times (446 MB per launch of the CLI's default set, for a 100 MB
// panorama).  So the tiles are grouped by the BLOCK of the source their footprint is centred in (Z-order cells,
// 512 x 256 pixels, smaller when that gives fewer than 64 groups), whole groups go to one XCD -- the tiles that share
// lines run at the same time on the same L2 -- and the groups are dealt heaviest first to the XCD with the least work
// so far (tiles whose footprint spans most of a row, next to a pole, count double).  CLI default set at 8K: 307 MB,
// 85 -> 73 us; blocks of 128 x 64 ... 256 x 128 pixels 79 / 75 us, 1024 x 512 83 us; one contiguous run of the order
// per XCD 104 us, contiguous source bands of equal cost per XCD (what serves the main kernel) 97-100 us: the polar
// tiles' cost is not a number the host can guess, and the XCDs that hold them are busy long after the others.
// by_source false: the tiles in list order, dealt round-robin (what the kernel's grid did before).
std::vector<uint32_t> xcd_lists(const std::vector<uint32_t>& tiles, const std::vector<p2p::PieceHdr>& hh, int pw, bool by_source,
                                int group_log2, int* stride)
{
    std::vector<std::vector<uint32_t>> per(8);
    if (!by_source) {
        for (size_t i = 0; i < tiles.size(); ++i)
            per[i & 7].push_back(tiles[i]);
    } else {
        std::vector<std::pair<uint64_t, uint32_t>> order;
        order.reserve(tiles.size());
        for (uint32_t s : tiles)
            order.emplace_back(source_order_key(hh[s]), s);
        std::sort(order.begin(), order.end());
        int g = group_log2;  // 2^g x 2^g cells (3: 512 x 256 source pixels)
        const size_t min_groups = 64;
        for (; g > 0; --g) {
            size_t groups = 0;
            for (size_t i = 0; i < order.size(); ++i)
                groups += i == 0 || (order[i].first >> (2 * g)) != (order[i - 1].first >> (2 * g));
            if (groups >= min_groups)
                break;
        }
        struct Group { size_t first, last; uint64_t key; long cost; };
        std::vector<Group> groups;
        for (size_t i = 0; i < order.size(); ++i) {
            const uint64_t k = order[i].first >> (2 * g);
            if (groups.empty() || k != groups.back().key)
                groups.push_back(Group{i, i, k, 0});
            groups.back().last = i + 1;
            const p2p::PieceHdr& h = hh[order[i].second];
            groups.back().cost += 1 + (h.c1 - h.c0 > pw / 2);
        }
        std::vector<size_t> by_cost(groups.size());
        for (size_t i = 0; i < by_cost.size(); ++i)
            by_cost[i] = i;
        std::stable_sort(by_cost.begin(), by_cost.end(), [&](size_t a, size_t b) { return groups[a].cost > groups[b].cost; });
        long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<std::vector<size_t>> mine(8);
        for (size_t gi : by_cost) {
            const int x = (int)(std::min_element(load, load + 8) - load);
            load[x] += groups[gi].cost;
            mine[x].push_back(gi);
        }
        for (int x = 0; x < 8; ++x) {
            std::sort(mine[x].begin(), mine[x].end());  // an XCD walks its groups in source order
            for (size_t gi : mine[x])
                for (size_t i = groups[gi].first; i < groups[gi].last; ++i)
                    per[x].push_back(order[i].second);
        }
    }
    size_t longest = 1;
    for (const auto& v : per)
        longest = std::max(longest, v.size());
    std::vector<uint32_t> table(8 * longest, ~0u);
    for (int x = 0; x < 8; ++x)
        std::copy(per[x].begin(), per[x].end(), table.begin() + x * longest);
    *stride = (int)longest;
    return table;
}


// tile shapes: 0 = 64 x 16 (LDS buffers of 704 items), 1 = 128 x 16 (1408), 2 = the band shape: 64 x 16 with buffers of 960
// items (p2p_views_band.hip) -- what a job drawn from source-band tiles gets unless P2P_TILE_SHAPE names one
const p2p::ShapeOps& shape_ops(int shape) { return shape == 2 ? p2p::shape_ops_w64b() : (shape ? p2p::shape_ops_w128() : p2p::shape_ops_w64()); }

// the source cell of a band plan, rows x columns: 16 x 8, 24 x 16 for the band shape's larger rectangles (CLI default set:
// 47.1 us with 16 x 8, 46.0 with 24 x 16), unless P2P_BAND_BH / P2P_BAND_CW say otherwise
void band_cell(const Options& o, int shape, int* bh, int* cw)
{
    *bh = o.band_bh > 0 ? o.band_bh : (shape == 2 ? 24 : 16);
    *cw = o.band_cw > 0 ? o.band_cw : (shape == 2 ? 16 : 8);
}

// Tile shape of a job (p2p_device.h: tile shapes): 128-wide tiles when the launch's views go well beyond the Infinity
// Cache and stream to HBM -- whole 128-byte lines per wave store -- (config 4: 18 GB, 6.5 ms against 7.2; config 3 on
// one GPU: 14 GB, 6.1 against 6.4), 64-wide tiles otherwise (config 2: 85 us against 95; config 5's 2.2 GB: 750
// against 768; the CLI's default set 73 against 93).
int choose_shape(const p2p_job_desc& d, const Options& opt)
{
    const int forced = opt.tile_shape;
    if (forced == 64 || forced == 128)
        return forced == 128;
    const size_t out_row = 12 * (((size_t)d.ow + 3) / 4);
    const size_t bytes = (size_t)d.n_panos * d.n_yaw * d.n_pitch * d.oh * out_row;
    // (several resident panoramas stream from HBM as well, and the 64-wide kernel's nt sc1 stores are for launches that
    // stay in the Infinity Cache: of config 2's panoramas 3, 0.67 GB of views, 253 us with 64-wide tiles against 267; 4,
    // 0.9 GB, 413 against 401; 8 0.83 against 0.79 ms; 16 1.64 against 1.56; ONE panorama and 2.2 GB, config 5, 720
    // against 768 us: tools/ab_shape_threshold.sh)
    const size_t from = d.n_panos > 1 ? (size_t)3 << 28 : (size_t)4 << 30;
    // (a strongly minifying view set is drawn by the gather kernel, which gains nothing from wide tiles: 16K -> 2048^2
    // at FOV 110, 4.5 GB, 5.28 ms with 64-wide tiles against 5.49)
    const double src_px_per_out_px = (double)d.pw * d.fov_deg / (360.0 * d.ow);
    return bytes >= from && d.ow >= 256 && src_px_per_out_px < 1.6;
}

int choose_pairs_per_block(const p2p_job_desc& d, const p2p::TileShape& S, const Options& opt)
{
    const int tiles = ((d.ow + S.tile_w - 1) / S.tile_w) * ((d.oh + S.tile_h - 1) / S.tile_h);
    const long long base = (long long)tiles * d.n_pitch;  // workgroups per pair chunk
    const int n_pairs = d.n_panos * d.n_yaw;
    int forced = opt.pairs_per_block;
    if (forced > 64)
        forced = 64;  // the kernel keeps one pair context per lane of a wave
    if (forced > 0)
        return forced > n_pairs ? n_pairs : forced;
    // about 8 workgroups per CU in flight, otherwise amortise the tile's set-up over many pairs.  Rounded to the nearest
    // count, not up: one pitch view of 1920 x 1080 is 2040 tiles, and two workgroups per tile instead of one cost a
    // 5-yaw job 22.4 us instead of 19.9, a 4-yaw job 20.1 instead of 17.1 (tools/ab_small_job_ppb.sh) -- the shares of
    // the view-sharded multi-GPU path
    const long long target = 256LL * 8;
    long long z = (target + base / 2) / base;
    if (z < 1) z = 1;
    if (z > n_pairs) z = n_pairs;
    int ppb = (int)((n_pairs + z - 1) / z);
    // (never fewer than 3 pairs behind one set-up, however few the tiles: 640 x 360, 230 tiles, 12 yaws: 1 / 2 / 3 / 4 / 6
    // pairs per workgroup 10.7 / 8.5 / 7.7 / 7.7 / 8.1 us)
    if (ppb < 3)
        ppb = n_pairs < 3 ? n_pairs : 3;
    // measured on the plan-driven kernel (config 5, 360 yaws): 16 pairs per workgroup 0.885 ms, 30: 0.843, 45: 0.835,
    // 60: 0.832 -- the per-workgroup set-up is small now.  Chunks that run across several panoramas are another
    // matter (8 resident panoramas: 16 pairs 0.843 ms, 48 pairs 0.894): their sources compete for the caches
    int cap = opt.max_pairs_per_block >= 0 ? opt.max_pairs_per_block : (d.n_panos > 1 ? 16 : 48);
    if (cap > 64) cap = 64;
    if (cap < 1) cap = 1;
    if (ppb > cap) {
        const int chunks = (n_pairs + cap - 1) / cap;  // even chunks instead of full ones plus a remainder
        ppb = (n_pairs + chunks - 1) / chunks;
    }
    // Several resident panoramas and no forced cap: whole panoramas per chunk -- about 20 pairs, a multiple of the yaw
    // count -- so that no workgroup's pairs straddle two panoramas (8 panoramas x 12 yaws: 24 pairs 792 us, 16 pairs
    // 803, 12 pairs 814: tools/ab_cfg3_share.sh)
    // (only up to 24 pairs -- what has been measured: 12 yaws at 12 / 16 / 24 pairs; 48 pairs across panoramas was slower
    // than 16 above, so a job of 33..64 yaws keeps the capped chunks)
    if (d.n_panos > 1 && opt.max_pairs_per_block < 0 && d.n_yaw <= 24) {
        const int per = std::max(1, (20 + d.n_yaw / 2) / d.n_yaw) * d.n_yaw;
        if (per <= 24 && per <= n_pairs)
            ppb = per;
    }
    return ppb < 1 ? 1 : ppb;
}

// bytes of plan tables one launch reads (per-pixel words and item lists of every tile of every pitch view)
size_t plan_table_bytes(const p2p_job_desc& d, const p2p::TileShape& S)
{
    const size_t tiles = (size_t)((d.ow + S.tile_w - 1) / S.tile_w) * ((d.oh + S.tile_h - 1) / S.tile_h);
    return tiles * (size_t)d.n_pitch * (S.block * S.pxt + S.cap) * sizeof(uint32_t);
}

// Chunks of pairs ONE main-kernel workgroup draws, one after the other.  1 as long as a launch's plan tables stay in the
// Infinity Cache: every chunk then has its own workgroup, and more of them are in flight.  Beyond that (config 4: 565 MB
// of tables per launch) every chunk's workgroup would pull the tile's words and items in again behind the launch's
// own write stream; one workgroup then draws all the chunks of its tile and reads them once.
int choose_main_span(const p2p_job_desc& d, const p2p::TileShape& S, const Options& opt, int pairs_per_block)
{
    const int chunks = (d.n_panos * d.n_yaw + pairs_per_block - 1) / pairs_per_block;
    int span = 1;
    if (S.tile_w != 128)
        return 1;  // (only the 128-wide kernel has the loop: p2p_views.hip, draw_tight)
    if (opt.main_span >= 1)
        span = opt.main_span;
    else if (d.n_panos == 1 && plan_table_bytes(d, S) > ((size_t)128 << 20))
        span = chunks;
    return std::max(1, std::min(span, chunks));
}

// List order: entries of an XCD's list that are drawn for one chunk of pairs before the next chunk (about the
// workgroups the XCD holds at a time: their tables and source rows are still in its L2 for the next chunk).  When one
// workgroup draws ALL the chunks of its tile the number only spaces the table-prefetch workgroups -- one per group,
// touching the tables of the group two further on: config 4 with 96 / 48 / 24 entries 6.39 / 6.37 / 6.34 ms.
int choose_main_group(const Options& opt, int shape, int span, int chunks)
{
    if (opt.main_group >= 0)
        return opt.main_group;
    if (span > 1 && span >= chunks)
        return 24;
    return shape == 1 ? 96 : 192;
}




// List order for the main kernel (xcd_main_lists): 1 = jobs with ONE resident panorama, 2 = also with several, 0 = the
// grid's own order.  By default: one panorama AND (several pitch views OR all pairs in one workgroup per tile).  The
// order exists to make views that read the same source rows neighbours on an XCD; a single pitch view drawn in several
// chunks of pairs gains nothing from it and pays its index arithmetic (config 5, 8 chunks: 0.742 ms in list order
// against 0.713).  With ONE chunk the lists' equal WORK per XCD (the grid gives every XCD the same number of tiles)
// and their split tail pay for a single pitch view too: 12 yaws of one 1080p view 37.5 -> 33.0 us, 5 yaws 19.4 -> 19.1
// (tools/ab_single_pitch_order.sh) -- the shares of the view-sharded path.
// Source-band tiles (p2p_device.h) instead of the main kernel's per-view tiles: where they apply at all -- the uint8
// path, BORDER_CONSTANT, every yaw a plain shift (no rest kernel), panorama width divisible by 4, the n_pitch views of a
// pair within one 32-bit descriptor -- and, unless P2P_BAND forces them, where they pay (choose_band's rule).
bool job_band_applies(const p2p_job* j)
{
    const p2p_job_desc& d = j->d;
    const p2p::TileShape& S = shape_ops(j->shape).shape;
    const Options& o = j->opt;
    if (d.flags & (P2P_FLAG_PIXELS_F32 | P2P_FLAG_PIXELS_F16))
        return false;
    if (j->border != 0 || j->n_odd_yaws > 0 || (d.pw & 3) != 0 || o.force_rest != 0)
        return false;
    if ((unsigned long long)d.n_pitch * d.oh * j->out_row >= (1ull << 32))
        return false;
    if ((unsigned long long)d.n_pitch * d.oh * ((d.ow + 3) / 4) >= (1ull << 32) - 1ull)
        return false;
    // a single cell must fit a tile's LDS buffer, whatever its groups look like
    int bh, cw;
    band_cell(o, j->shape, &bh, &cw);
    const long rows = bh + o.band_maxh + 1, ri = ((cw + o.band_maxw + 3) >> 2) + 1;
    if (rows * ri > S.cap || rows > 65535)
        return false;
    if ((d.pw + cw - 1) / cw > 4096)  // (p2p_plan.hip: BAND_MAX_NCX, the cut's cells in LDS)
        return false;
    return true;
}

bool job_wants_band(const p2p_job* j)
{
    if (j->opt.band == 0 || !job_band_applies(j))
        return false;
    if (j->opt.band > 0)
        return true;
    // The library's rule (tools/band_rule.py, profiles/r05_band_rule_sweep.txt).  Band tiles do MORE arithmetic per
    // output pixel than the gather kernel (stage 1 for the whole source rectangle, not for the taps alone) and win where
    // that kernel waits for its scattered lines: view sets that minify enough for the per-view tiles' footprints to
    // overflow the LDS (about 1.25 source pixels per output pixel) but not so much that the rectangle is mostly gaps
    // (3.2), with enough (panorama, yaw) pairs per tile to pay its set-up -- 8, or 4 when five pitch views share every
    // source rectangle, as in the reference CLI's defaults -- and views small enough for the write-back stores their
    // ragged edges need (p2p_tile.h: P2P_BAND_STORE_AUX): 256 MB per launch.  Config 2 (1.07) stays with the per-view tiles.
    const p2p_job_desc& d = j->d;
    const double r = (double)d.pw * j->fov / (360.0 * d.ow);
    const long long pairs = (long long)d.n_panos * d.n_yaw;
    // (a panorama beyond the Infinity Cache, 16K: the gather kernel's scattered lines come from HBM, and band tiles win up
    // to 4 source pixels per output pixel -- 12 x 3 views of 1024 x 576 146 us against 180, 4 x 5 of 1024^2 181 against 191)
    const double r_max = (size_t)d.pw * d.ph * 3 > ((size_t)256 << 20) ? 4.2 : 3.2;
    if (r < 1.25 || r > r_max || j->d_view_mask || j->host_maps)  // (caller maps: their minification is not the FOV's)
        return false;
    if (!(pairs >= 8 || (pairs >= 4 && d.n_pitch >= 5)))
        return false;
    // (with fewer than 8 pairs per tile the bound is 80 MB: 4 yaws x 5 pitches of 1152 x 1152 from 8K, 76 MB, 63.5 us
    // against 71.6; of 1280 x 1280, 94 MB, 80.4 against 73.2 -- profiles/r05_band_rule_sweep.txt)
    // (with 8 pairs and more no row of the sweep loses by size: 8K 12 x 3 of 1536 x 864, 143 MB, 69.7 us against 76.9; 16K
    // 12 x 3 of 2048 x 1152, 255 MB, 225 against 256; 398 MB -6 %, 573 MB -1 %)
    return j->out_bytes <= ((size_t)(pairs >= 8 ? 256 : 80) << 20);
}

// A job that is drawn from source-band tiles gets the band shape (unless P2P_TILE_SHAPE names one); its yaws may change
// (p2p_job_set_yaws, p2p_job_set_maps: an odd yaw takes the band plan away), so the shape is settled again before every
// plan look-up.  Shapes 0 and 2 share the tile raster: nothing else of the job depends on which of the two it is.
void job_settle_shape(p2p_job* j)
{
    const int base = choose_shape(j->d, j->opt);
    if (base != 0 || j->opt.tile_shape == 64 || j->opt.tile_shape == 128) {
        j->shape = base;
        return;
    }
    j->shape = 2;
    if (!job_wants_band(j))
        j->shape = 0;
}

int job_main_order(const p2p_job* j)
{
    if (j->opt.main_order >= 0)
        return j->opt.main_order;
    if (j->d.n_panos != 1)
        return 0;
    if (j->d.n_pitch > 1)
        return 1;
    const int ppb = choose_pairs_per_block(j->d, shape_ops(j->shape).shape, j->opt);
    return (j->d.n_yaw + ppb - 1) / ppb == 1 ? 1 : 0;
}


// the ONE block job_build_plan allocates for a plan's tables, carved at 256-byte boundaries
struct PlanBlock {
    size_t b_px2, b_coords, b_hdr, b_px, b_items, b_list;
    size_t total() const { return b_px2 + b_coords + b_hdr + b_px + b_items + b_list; }
};
static PlanBlock plan_block_layout(const p2p_job_desc& d, const p2p::TileShape& S, size_t slots, bool band, bool float_path)
{
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    PlanBlock L;
    L.b_px2 = float_path ? up(slots * S.block * S.pxt * sizeof(uint32_t)) : 0;
    L.b_coords = up((size_t)d.n_pitch * d.oh * d.ow * sizeof(int2));
    L.b_hdr = up(slots * sizeof(p2p::PieceHdr));
    L.b_px = band ? 0 : up(slots * S.block * S.pxt * sizeof(uint32_t));
    L.b_items = band ? 0 : up(slots * S.cap * sizeof(uint32_t));
    L.b_list = up(slots * sizeof(uint32_t));
    return L;
}

// the key a job's plan is kept under in its context (PlanKey: the reference's cache key for the whole pitch list, plus what
// shapes the tables)
static PlanKey plan_key_of(const p2p_job* j, bool band, int cell_bh, int cell_cw, int main_order)
{
    const p2p_job_desc& d = j->d;
    const Options& opt = j->opt;
    return PlanKey{d.pw, d.ph, d.ow, d.oh, d.flags & (P2P_FLAG_PIXELS_F32 | P2P_FLAG_PIXELS_F16 | P2P_FLAG_PIXEL_CENTRES), j->border,
                   j->fov, j->pitch, j->shape, {opt.gather_blocky_from, main_order != 0, opt.gather_order, opt.gather_group,
                                                band ? 1 : 0, band ? cell_bh : 0, band ? cell_cw : 0, band ? opt.band_maxw : 0, band ? opt.band_maxh : 0,
                                                j->row0, j->row1}};  // (the rows the job draws: p2p_job_set_rows)
}

// p2p_job_create, while the device makes the job's yaw tables: the block the job's first p2p_job_run will ask the pool for
// is fetched from the driver now and handed to the pool (hipMalloc is 12 us whatever the size: a fresh geometry's first
// image no longer waits for it).  A prediction -- the plan the job would build if it ran now; a job that changes its mind
// (p2p_job_set_maps, a view mask) finds the pool without that block, as before.  Nothing when the context has the plan.
void plan_block_prefetch(p2p_job* j) noexcept
{
    try {
        if (options().pool_mb <= 0)
            return;
        job_settle_shape(j);
        const bool band = job_wants_band(j);
        int cell_bh, cell_cw;
        band_cell(j->opt, j->shape, &cell_bh, &cell_cw);
        const int main_order = band ? 0 : job_main_order(j);
        if (j->opt.plan_cache != 0) {
            const PlanKey key = plan_key_of(j, band, cell_bh, cell_cw, main_order);
            std::lock_guard<std::mutex> lk(j->ctx->cache_mu);
            if (j->ctx->plans.find(key) != j->ctx->plans.end())
                return;
        }
        const bool float_path = (j->d.flags & (P2P_FLAG_PIXELS_F32 | P2P_FLAG_PIXELS_F16)) != 0;
        const PlanBlock L = plan_block_layout(j->d, shape_ops(j->shape).shape, j->n_tiles * (size_t)j->d.n_pitch, band, float_path);
        void* blk = nullptr;
        if (dev_alloc(&blk, L.total()) == hipSuccess)
            (void)dev_free(blk);
        else
            (void)hipGetLastError();
    } catch (...) {
    }
}

// Build the job's plan (p2p_plan.hip) for its current maps: once per job geometry, like the yaw tables.  It
// depends on the maps only, never on pixel data -- the device counterpart of the reference's
// pitch_mapping_cache (P:17-18, P:55-73), which lives as long as the process.
// after_plan_pass: called once the plan pass is enqueued and before anything waits for it (per-view plans only) --
// p2p_job_run launches the main kernel in grid order there, so that one image through a fresh context has its pixels
// under way while the host reads the gather count back.
int job_build_plan(p2p_job* j, const std::function<int(Plan&)>& after_plan_pass)
{
    const p2p_job_desc& d = j->d;
    p2p_ctx* ctx = j->ctx;
    hipStream_t st = ctx->stream;
    const bool float_path = (d.flags & (P2P_FLAG_PIXELS_F32 | P2P_FLAG_PIXELS_F16)) != 0;
    const size_t slots = j->n_tiles * d.n_pitch;
    // device maps: the plan is a function of the key alone -- the context may have it already
    const Options& opt = j->opt;
    const bool band = job_wants_band(j);
    int cell_bh, cell_cw;
    band_cell(opt, j->shape, &cell_bh, &cell_cw);
    const int main_order = band ? 0 : job_main_order(j);
    const PlanKey key = plan_key_of(j, band, cell_bh, cell_cw, main_order);
    const p2p::TileShape& S = shape_ops(j->shape).shape;
    const bool cached = !j->host_maps && opt.plan_cache != 0 && opt.scramble_plan == 0;
    if (cached) {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        auto it = ctx->plans.find(key);
        if (it != ctx->plans.end() && it->second->built) {
            it->second->stamp = ++ctx->cache_clock;
            j->plan_ref = it->second;
            return P2P_OK;
        }
    }
    auto Pl = std::make_shared<Plan>();
    Pl->device = ctx->device;
    // host buffers that asynchronous copies read from or write to: declared BEFORE the guard, so that on every error
    // return the stream is drained (the guard, destroyed first) before they and the plan's blocks -- back to the pool,
    // where any thread may pick them up at once -- go out of scope
    std::vector<p2p::PieceHdr> hh;
    uint32_t cnt = 0;
    p2p::BandInfo binfo{};
    // the read-backs land in a pinned block first (pin_get): [BandInfo | counter | headers]
    PinnedBlock pin;
    const size_t pin_hdr_off = 128, pin_hdr_max = 65536;  // (larger plans read their headers straight into hh)
    // the band passes' scratch (cells, the groups' cells, the cut's records, the sorted group list): freed on every
    // path, after the stream has been drained (declared before the guard: destroyed after it)
    struct Scratch {
        std::vector<void*> blocks;
        size_t total = 0;
        ~Scratch() { for (void* b : blocks) (void)dev_free(b); }
        hipError_t get(void** out, size_t bytes)
        {
            blocks.reserve(blocks.size() + 1);  // (first: a block that cannot be listed would never be freed)
            hipError_t e = dev_alloc(out, bytes);
            if (e == hipSuccess) {
                blocks.push_back(*out);
                total += bytes;
            }
            return e;
        }
    } scratch;
    std::unique_lock<std::mutex> one_pass(ctx->plan_mu, std::defer_lock);  // (declared first: given back after the stream has drained)
    StreamSyncGuard sync_on_exit(st);
    HIP_TRY(pin_get(&pin.p, &pin.cls, pin_hdr_off + std::min(slots, pin_hdr_max) * sizeof(p2p::PieceHdr)));
    p2p::BandInfo* const h_binfo = (p2p::BandInfo*)pin.p;
    uint32_t* const h_cnt = (uint32_t*)((unsigned char*)pin.p + 112);
    p2p::PieceHdr* const h_hdr = (p2p::PieceHdr*)((unsigned char*)pin.p + pin_hdr_off);
    static_assert(sizeof(p2p::BandInfo) <= 96, "the pinned block's layout: [BandInfo 0..96 | band_scan_kernel's words 96..112 | counter 112 | headers 128..]");
    // the headers into hh: through the pinned block (one asynchronous copy, unpacked after the stream's next
    // synchronisation by hdr_arrived) or, beyond its size, straight into the vector
    bool hdr_in_pin = false;
    auto fetch_headers = [&]() -> hipError_t {
        hh.resize(slots);
        hdr_in_pin = slots <= pin_hdr_max;
        return hipMemcpyAsync(hdr_in_pin ? (void*)h_hdr : (void*)hh.data(), Pl->d_hdr, slots * sizeof(p2p::PieceHdr), hipMemcpyDeviceToHost, st);
    };
    auto hdr_arrived = [&]() {
        if (hdr_in_pin)
            memcpy(hh.data(), h_hdr, slots * sizeof(p2p::PieceHdr));
        hdr_in_pin = false;
    };
    Pl->band = band;
    {
        // ONE block for the plan pass's tables (the device is idle while a fresh geometry's blocks are mapped: six
        // allocations were a third of the first image's device-side time), carved at 256-byte boundaries
        const PlanBlock L = plan_block_layout(d, S, slots, band, float_path);
        const size_t b_px2 = L.b_px2, b_coords = L.b_coords, b_hdr = L.b_hdr, b_px = L.b_px, b_items = L.b_items, b_list = L.b_list;
        unsigned char* blk = nullptr;
        HIP_TRY(dev_alloc((void**)&blk, L.total()));
        Pl->d_block = blk;
        size_t off = 0;
        auto take = [&](size_t b) { unsigned char* p = b ? blk + off : nullptr; off += b; return p; };
        Pl->d_px2 = (uint32_t*)take(b_px2);
        Pl->d_coords = (int2*)take(b_coords);
        Pl->d_hdr = (p2p::PieceHdr*)take(b_hdr);
        Pl->d_px = (uint32_t*)take(b_px);
        Pl->d_items = (uint32_t*)take(b_items);
        Pl->d_gather_list = (uint32_t*)take(b_list);
    }
    Pl->bytes = (size_t)d.n_pitch * d.oh * d.ow * sizeof(int2) +
                slots * (sizeof(p2p::PieceHdr) + ((band ? 0 : S.block * S.pxt * (float_path ? 2 : 1) + S.cap) + 1) * sizeof(uint32_t));
    p2p::PlanParams Q{};
    Q.pw = d.pw; Q.ph = d.ph; Q.ow = d.ow; Q.oh = d.oh; Q.n_pitch = d.n_pitch; Q.border = j->border;
    Q.geom = j->geom;
    Q.pitch = j->d_pitch;
    Q.mapU = j->host_maps ? j->d_mapU : nullptr;
    Q.mapV = j->host_maps ? j->d_mapV : nullptr;
    Q.coords = Pl->d_coords;
    Q.ty0 = j->row0 / S.tile_h;
    Q.ty1 = (j->row1 + S.tile_h - 1) / S.tile_h;
    Q.coords_all = (band || float_path || opt.coords_all != 0) ? 1 : 0;
    Pl->coords_full = Q.coords_all != 0;
    Q.hdr = Pl->d_hdr;
    Q.px = Pl->d_px;
    Q.items = Pl->d_items;
    Q.blocky_from = opt.gather_blocky_from;
    Q.gather_list = Pl->d_gather_list;
    Q.float_path = float_path;
    Q.centre = (d.flags & P2P_FLAG_PIXEL_CENTRES) ? 0.5f : 0.0f;
    Q.px2 = Pl->d_px2;
    p2p::BandParams& B = Q.band;
    uint32_t* d_cnt = nullptr;  // the gather tiles' counter: the context's (per-view plans) or inside the cell block (band plans), see below
    const size_t n_groups_all = (size_t)d.n_pitch * d.oh * ((d.ow + 3) / 4);
    if (band) {
        B.pw = d.pw; B.ph = d.ph; B.ow = d.ow; B.oh = d.oh; B.n_pitch = d.n_pitch;
        B.g.bh = cell_bh; B.g.cw = cell_cw;
        B.g.ncx = (d.pw + B.g.cw - 1) / B.g.cw;
        B.g.n_bands = (d.ph + B.g.bh - 1) / B.g.bh;
        B.g.maxw = opt.band_maxw; B.g.maxh = opt.band_maxh;
        B.coords = Pl->d_coords;
        B.out_row = j->out_row;
        B.view_bytes = (size_t)d.oh * j->out_row;
        B.cost_base = S.tile_w == 128 ? 675u : 600u;  // (xcd_main_lists' cost model)
        const size_t cells = (size_t)B.g.n_bands * B.g.ncx;
        // count, cmax1, rmax1, cur, cmin (kept as INT32_MAX - column: zero = no group yet), the plan pass's gather counter
        // -- ONE memset zeroes them all (four were 40 us of a cold image's 415: a launch and its gap each) --, then off
        uint32_t* cellblk = nullptr;
        HIP_TRY(scratch.get((void**)&cellblk, (cells * 6 + 64) * sizeof(uint32_t)));
        B.cell_count = cellblk; B.cell_cmax1 = (int*)(cellblk + cells); B.cell_rmax1 = (int*)(cellblk + 2 * cells);
        B.cell_cur = cellblk + 3 * cells; B.cell_cmin = (int*)(cellblk + 4 * cells); B.cell_off = cellblk + 5 * cells + 64;
        d_cnt = cellblk + 5 * cells;
        Q.n_gather = d_cnt;
        HIP_TRY(scratch.get((void**)&B.gcell, n_groups_all * sizeof(uint32_t)));
        HIP_TRY(scratch.get((void**)&B.band_cost, (size_t)B.g.n_bands * (sizeof(unsigned long long) + 2 * sizeof(uint32_t))));
        B.band_tiles = (uint32_t*)(B.band_cost + B.g.n_bands);
        B.band_groups = B.band_tiles + B.g.n_bands;
        HIP_TRY(dev_alloc((void**)&Pl->d_band_info, sizeof(p2p::BandInfo)));
        B.info = Pl->d_band_info;
        // (BandInfo: every field is written by band_scan_kernel / band_xcd_kernel)
        HIP_TRY(p2p::launch_zero_words(cellblk, (uint32_t)(cells * 5 + 64), st));
    }
#ifdef P2P_AUDIT
    if (!band) {
        // every pool poisoned: a kernel that reads a slot the plan pass did not write gets 0xFF.. and the audit sees it
        HIP_TRY(hipMemsetAsync(Pl->d_hdr, 0xFF, slots * sizeof(p2p::PieceHdr), st));
        HIP_TRY(hipMemsetAsync(Pl->d_px, 0xFF, slots * S.block * S.pxt * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(Pl->d_items, 0xFF, slots * S.cap * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(Pl->d_gather_list, 0xFF, slots * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(Pl->d_coords, 0xFF, (size_t)d.n_pitch * d.oh * d.ow * sizeof(int2), st));
        if (Pl->d_px2)
            HIP_TRY(hipMemsetAsync(Pl->d_px2, 0xFF, slots * S.block * S.pxt * sizeof(uint32_t), st));
    }
#endif
    // (the plan pass writes every header, every per-pixel word and every item slot of every tile: nothing to clear)
    // Per-view plans: the plan pass's last workgroup writes the gather count into the page-locked block itself
    // (plan_kernel), and the host waits for the pass's event alone -- no copy in the stream behind the main kernel that is
    // about to follow, no waiting for that kernel.  The pass counts in the context's own words (zero between passes: no
    // clearing in front of it either), so one pass per context at a time.
    constexpr uint32_t kNoCount = 0xFFFFFFFFu;
    // Spin until the device has written a word of the page-locked block (it starts as kNoCount); now and then: has the
    // stream drained, or failed, without it?  P2P_OK also when it drained without the word: the caller looks at the word.
    auto wait_for_word = [&](volatile uint32_t* w, const char* what) -> int {
        for (unsigned spins = 1; *w == kNoCount; ++spins) {
            if ((spins & 2047u) == 0u) {
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess)
                    break;
                if (q != hipErrorNotReady)
                    return fail(P2P_ERR_HIP, "%s: %s", what, hipGetErrorString(q));
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        return P2P_OK;
    };
    volatile uint32_t* const h_words = (volatile uint32_t*)((unsigned char*)pin.p + 96);  // (band plans: band_scan_kernel's four words)
    if (band) {
        h_words[0] = h_words[1] = h_words[2] = 0u;
        h_words[3] = kNoCount;
        B.host_words = (uint32_t*)h_words;
        B.n_gather = d_cnt;
        if (slots <= pin_hdr_max) {  // the headers straight into the page-locked block (hdr_arrived unpacks them)
            Q.hdr_host = h_hdr;
            hh.resize(slots);
            hdr_in_pin = true;
        }
    }
    if (!band) {
        one_pass.lock();
        Q.n_gather = d_cnt = ctx->d_plan_cnt + p2p::PLAN_TICKET_WORDS;
        Q.ticket = ctx->d_plan_cnt;
        Q.n_gather_host = h_cnt;
        *(volatile uint32_t*)h_cnt = kNoCount;
    }
    // (the pass -- a band plan's passes -- is timed only for a job that asked for launch timing: an event in front of the
    // pass and one between it and the main kernel are 5 us each on a cold image's critical path, and a band plan's time
    // could only be read after waiting for the whole stream)
    const bool timed_plan = j->time_launches;
    if (timed_plan)
        HIP_TRY(hipEventRecord(ctx->ev_t0, st));
    HIP_TRY(shape_ops(j->shape).plan(Q, st));
    if (!band) {
        if (timed_plan)
            HIP_TRY(hipEventRecord(ctx->ev_t1, st));
        if (after_plan_pass)
            if (int rc = after_plan_pass(*Pl))
                return rc;
    }
    if (band) {
        // the band passes: count the tiles, hand the counts to the host (the tables are sized by them), cut, sort, build.
        // band_scan_kernel writes the three counts into the page-locked block itself and the plan pass has written every
        // header there (plans of up to 65536 tiles): no copy in the stream, and the host waits for one word -- it makes the
        // gather tiles' lists from the headers WHILE the device cuts, sorts and builds the band tiles.
        HIP_TRY(shape_ops(j->shape).band(B, 0, st));
        if (int rc = wait_for_word(h_words + 3, "the band passes"))
            return rc;
        if (h_words[3] != 1u) {  // (the stream drained without the flag: read the counts back the old way)
            HIP_TRY(hipMemcpyAsync(h_binfo, B.info, sizeof(binfo), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            binfo = *h_binfo;
            cnt = *h_cnt;
        } else {
            binfo.n_tiles = h_words[0];
            binfo.n_groups = h_words[1];
            cnt = h_words[2];
        }
        if (hdr_in_pin)  // (the plan pass wrote them there, two kernels ago)
            hdr_arrived();
        if ((size_t)binfo.n_groups > n_groups_all || (size_t)binfo.n_tiles > n_groups_all)
            return fail(P2P_ERR_HIP, "the band passes counted %u tiles, %u groups of %zu", binfo.n_tiles, binfo.n_groups, n_groups_all);
        B.n_tiles = (int)binfo.n_tiles;
        B.n_groups = (int)binfo.n_groups;
        Pl->band_tiles = B.n_tiles;
        Pl->band_groups = B.n_groups;
        // list entries per XCD the view kernel's grid provides for: a quarter more than an equal share (the runs are of
        // equal WORK; band_xcd_kernel keeps every run within this)
        Pl->band_per = (B.n_tiles + 7) / 8 + (B.n_tiles + 31) / 32 + 16;
        B.per_cap = Pl->band_per;
        if (B.n_tiles > 0) {
            const size_t nt = (size_t)B.n_tiles;
            HIP_TRY(scratch.get((void**)&B.recs, nt * sizeof(p2p::BandTileRec)));
            HIP_TRY(scratch.get((void**)&B.sorted, (size_t)std::max(1, B.n_groups) * sizeof(uint32_t)));
            HIP_TRY(dev_alloc((void**)&Pl->d_band_hdr, nt * sizeof(p2p::PieceHdr)));
            HIP_TRY(dev_alloc((void**)&Pl->d_band_px, nt * S.block * S.pxt * sizeof(uint32_t)));
            HIP_TRY(dev_alloc((void**)&Pl->d_band_grp, nt * S.block * sizeof(uint32_t)));
            Pl->bytes += nt * (sizeof(p2p::PieceHdr) + (S.block * S.pxt + S.block) * sizeof(uint32_t));
            B.hdr = Pl->d_band_hdr; B.px = Pl->d_band_px; B.grp = Pl->d_band_grp;
            HIP_TRY(shape_ops(j->shape).band(B, 1, st));
        }
    }
    if (band && timed_plan)
        HIP_TRY(hipEventRecord(ctx->ev_t1, st));
    // the work lists are made from the plan's headers, once per geometry: they come back with the counter -- unless the
    // plan turns out to have no gather tile and may draw its first launch in grid order (Plan::lists_pending)
    const bool want_main_order = main_order != 0;
    const bool may_defer = want_main_order && opt.defer_lists == 2 && opt.main_order < 0 && opt.scramble_plan == 0;
    Pl->tile_w = shape_ops(j->shape).shape.tile_w;
    if (!band) {  // (band plans: the counter came back with the band counts; the device is still building the tiles)
        // the plan pass's last workgroup writes the word; whatever is queued behind the pass is not waited for
        if (int rc = wait_for_word((volatile uint32_t*)h_cnt, "the plan pass"))
            return rc;
        cnt = *(volatile uint32_t*)h_cnt;
        if (cnt == kNoCount || (size_t)cnt > slots) {
            // The stream has drained without a (credible) word: the context's counters were not zero when the pass began
            // (a lost write: DESIGN.md section 5.7).  The headers say which tiles gather; the counters are cleared again and the
            // plan does without its unordered gather list (the per-XCD lists are made from the headers anyway).
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(p2p::launch_zero_words(ctx->d_plan_cnt, p2p::PLAN_TICKET_WORDS + 32, st));
            HIP_TRY(fetch_headers());
            HIP_TRY(hipStreamSynchronize(st));
            hdr_arrived();
            std::vector<uint32_t> marked;
            for (size_t s = 0; s < slots; ++s)
                if ((hh[s].mode_items & 3u) == 2u)
                    marked.push_back((uint32_t)s);
            cnt = (uint32_t)marked.size();
            if (cnt > 0)
                HIP_TRY(hipMemcpy(Pl->d_gather_list, marked.data(), marked.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        Pl->plan_ms = 0.0f;
        if (timed_plan) {
            HIP_TRY(hipEventSynchronize(ctx->ev_t1));
            (void)hipEventElapsedTime(&Pl->plan_ms, ctx->ev_t0, ctx->ev_t1);
        }
        one_pass.unlock();
    }
    if ((size_t)cnt > slots)
        return fail(P2P_ERR_HIP, "the plan pass listed %u gather tiles of %zu", cnt, slots);
    Pl->n_gather = (int)cnt;
    Pl->built = true;
    bool make_main_list = want_main_order && (size_t)cnt < slots && !Pl->lists_made;  // (made already: in front of the first main kernel)
    if (make_main_list && may_defer && cnt == 0) {
        Pl->lists_pending = true;
        make_main_list = false;
    }
    if (cnt > 0 && hh.empty()) {  // (the gather tiles' lists are made on the host, from the headers)
        HIP_TRY(fetch_headers());
        HIP_TRY(hipStreamSynchronize(st));
        hdr_arrived();
    }
    if (make_main_list) {  // (on the device: one launch, nothing waited for)
        if (int rc = plan_enqueue_main_lists(*Pl, slots, shape_ops(j->shape).shape.tile_w, st))
            return rc;
        Pl->lists_made = true;
    }
    if (cnt > 0) {
        // the gather kernel's work lists, one per XCD: xcd_lists
        const bool by_source = opt.gather_order != 0;
        std::vector<uint32_t> marked, all;
        for (size_t s = 0; s < slots; ++s)
            if ((hh[s].mode_items & 3u) == 2u)
                marked.push_back((uint32_t)s);
        if (marked.size() != (size_t)cnt) {
            return fail(P2P_ERR_HIP, "the plan's headers mark %zu gather tiles, its counter %u", marked.size(), cnt);
        }
        std::vector<uint32_t>& tg = Pl->h_xcd_list;  // (kept with the plan: the upload below is not waited for)
        tg = xcd_lists(marked, hh, d.pw, by_source, opt.gather_group, &Pl->xcd_stride);
        HIP_TRY(dev_alloc((void**)&Pl->d_xcd_list, tg.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(Pl->d_xcd_list, tg.data(), tg.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        Pl->bytes += tg.size() * sizeof(uint32_t);
        // (not for a job that draws a range of rows: "every tile" would be the other ranks' too)
        if ((slots - (size_t)cnt) * 4 <= slots && j->row0 == 0 && j->row1 == d.oh) {  // the gather kernel may draw every tile (p2p_job_run: gather_all)
            all.resize(slots);
            for (size_t s = 0; s < slots; ++s)
                all[s] = (uint32_t)s;
            std::vector<uint32_t>& ta = Pl->h_xcd_all;
            ta = xcd_lists(all, hh, d.pw, by_source, opt.gather_group, &Pl->xcd_all_stride);
            HIP_TRY(dev_alloc((void**)&Pl->d_xcd_all, ta.size() * sizeof(uint32_t)));
            HIP_TRY(hipMemcpyAsync(Pl->d_xcd_all, ta.data(), ta.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            Pl->bytes += ta.size() * sizeof(uint32_t);
        }
    }
    if (band) {
        // (nothing is waited for: the band tiles are still being built when p2p_job_run enqueues the view kernel behind them;
        // the passes' scratch goes with the plan)
        Pl->plan_ms = 0.0f;
        if (timed_plan) {
            HIP_TRY(hipStreamSynchronize(st));
            (void)hipEventElapsedTime(&Pl->plan_ms, ctx->ev_t0, ctx->ev_t1);
        }
    }
    if (const int seed = opt.scramble_plan) {
        // Robustness self-test (tests/fuzz/scramble_tables.py), never set in normal use: every table of the plan -- and
        // with bit 30 of the value the job's yaw tables too -- overwritten with pseudo-random words AFTER the plan pass.
        // The view kernels must then draw garbage and nothing worse: every table-derived offset is clamped or goes
        // through an exact-extent buffer descriptor (csrc/p2p_audit.h).  Such a plan is never entered in the cache.
        const uint32_t sd = (uint32_t)seed;
        HIP_TRY(p2p::launch_scramble(Pl->d_hdr, slots * sizeof(p2p::PieceHdr), sd + 1, st));
        HIP_TRY(p2p::launch_scramble(Pl->d_px, Pl->d_px ? slots * S.block * S.pxt * sizeof(uint32_t) : 0, sd + 2, st));
        HIP_TRY(p2p::launch_scramble(Pl->d_items, Pl->d_items ? slots * S.cap * sizeof(uint32_t) : 0, sd + 3, st));
        if (Pl->band_tiles > 0) {
            const size_t nt = (size_t)Pl->band_tiles;
            HIP_TRY(p2p::launch_scramble(Pl->d_band_hdr, nt * sizeof(p2p::PieceHdr), sd + 13, st));
            HIP_TRY(p2p::launch_scramble(Pl->d_band_px, nt * S.block * S.pxt * sizeof(uint32_t), sd + 14, st));
            HIP_TRY(p2p::launch_scramble(Pl->d_band_grp, nt * S.block * sizeof(uint32_t), sd + 15, st));
            HIP_TRY(p2p::launch_scramble(Pl->d_band_info, sizeof(p2p::BandInfo), sd + 16, st));
        }
        HIP_TRY(p2p::launch_scramble(Pl->d_gather_list, slots * sizeof(uint32_t), sd + 4, st));
        HIP_TRY(p2p::launch_scramble(Pl->d_xcd_list, Pl->d_xcd_list ? 8 * (size_t)Pl->xcd_stride * sizeof(uint32_t) : 0, sd + 10, st));
        HIP_TRY(p2p::launch_scramble(Pl->d_main_list, Pl->d_main_list ? 8 * (size_t)Pl->main_stride * sizeof(uint32_t) : 0, sd + 12, st));
        HIP_TRY(p2p::launch_scramble(Pl->d_main_count, Pl->d_main_count ? 8 * sizeof(uint32_t) : 0, sd + 17, st));
        HIP_TRY(p2p::launch_scramble(Pl->d_xcd_all, Pl->d_xcd_all ? 8 * (size_t)Pl->xcd_all_stride * sizeof(uint32_t) : 0, sd + 11, st));
        HIP_TRY(p2p::launch_scramble(Pl->d_coords, (size_t)d.n_pitch * d.oh * d.ow * sizeof(int2), sd + 5, st));
        if (Pl->d_px2)
            HIP_TRY(p2p::launch_scramble(Pl->d_px2, slots * S.block * S.pxt * sizeof(uint32_t), sd + 6, st));
        if ((seed & (1 << 30)) && j->yaw_ref && j->yaw_ref.use_count() == 1) {  // private (uncached) yaw tables only
            const size_t n = (size_t)d.n_yaw * d.pw * sizeof(uint32_t);
            HIP_TRY(p2p::launch_scramble(j->d_ytab, n, sd + 7, st));
            HIP_TRY(p2p::launch_scramble(j->d_f4tab, n, sd + 8, st));
            HIP_TRY(p2p::launch_scramble(j->d_ydesc, (size_t)d.n_yaw * sizeof(p2p::YawDesc), sd + 9, st));
        }
        if (seed & (1 << 29))
            Pl->n_gather = (int)slots;  // every list entry is launched: the scrambled ones too
        HIP_TRY(hipStreamSynchronize(st));
        sync_on_exit.armed = false;
        j->plan_ref = Pl;
        return P2P_OK;
    }
    // every copy into host memory above has been waited for; kernels and uploads may still be queued: what they use stays
    // with the plan
    Pl->build_blocks.reserve(scratch.blocks.size());
    Pl->build_blocks.swap(scratch.blocks);
    Pl->bytes += scratch.total;
    sync_on_exit.armed = false;
    if (cached) {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        Pl->stamp = ++ctx->cache_clock;
        ctx->plans[key] = Pl;
        cache_trim(ctx);
    }
    j->plan_ref = Pl;
    return P2P_OK;
}


// Every pixel's quantised coordinates into a plan that kept only its gather tiles' (PlanParams::coords_all): for
// p2p_job_get_coords, for a gather kernel that is about to draw every tile, for yaw rows that are not a shift.
int ensure_full_coords(p2p_job* j)
{
    Plan& Pl = *j->plan_ref;
    std::lock_guard<std::mutex> lk(Pl.lists_mu);
    if (Pl.coords_full)
        return P2P_OK;
    p2p::PlanParams Q{};
    Q.pw = j->d.pw; Q.ph = j->d.ph; Q.ow = j->d.ow; Q.oh = j->d.oh; Q.n_pitch = j->d.n_pitch; Q.border = j->border;
    Q.geom = j->geom;
    Q.pitch = j->d_pitch;
    Q.mapU = j->host_maps ? j->d_mapU : nullptr;
    Q.mapV = j->host_maps ? j->d_mapV : nullptr;
    Q.coords = Pl.d_coords;
    Q.coords_only = 1;
    HIP_TRY(shape_ops(j->shape).plan(Q, j->ctx->stream));
    Pl.coords_full = true;
    return P2P_OK;
}

// the deferred half of job_build_plan: the main kernel's per-XCD lists of a plan that has been launched once -- one
// kernel on the job's stream in front of the launch that asked (round 5: a read-back, a host sort and an upload, 75 us of
// a geometry's second launch)
int plan_make_main_lists(p2p_job* j, Plan& Pl)
{
    std::lock_guard<std::mutex> lk(Pl.lists_mu);
    if (!Pl.lists_pending)
        return P2P_OK;
    if (int rc = plan_enqueue_main_lists(Pl, j->n_tiles * (size_t)j->d.n_pitch, Pl.tile_w, j->ctx->stream))
        return rc;
    Pl.lists_pending = false;
    return P2P_OK;
}

}  // namespace p2p_host
