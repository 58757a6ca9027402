// p2p_host.h -- what the host-side units of libp2p_hip.so share (NOT part of the C ABI: that is include/p2p_hip.h).
//
//   p2p_abi.cpp           every extern "C" entry point: argument hand-over + the exception barrier, nothing else
//   p2p_host_pool.cpp     error text, the P2P_* options, the device memory pool, the pinned read-back blocks
//   p2p_host_ctx.cpp      contexts (stream, events), their geometry-keyed table caches, the yaw tables
//   p2p_host_plan.cpp     tile shape / chunking rules, the plan pass and band passes' host side, the per-XCD work lists
//   p2p_host_job.cpp      jobs: buffers, uploads, p2p_job_run's launch sequence, downloads, timing
//   p2p_host_oneshot.cpp  the one-shot slot pool and the host-buffer entry points built on it
//
// Everything lives in namespace p2p_host; the implementation of the ABI function p2p_xyz is p2p_host::xyz.  No pixel or
// map arithmetic happens on the CPU here; the only host maths is the handful of float64 scalars NumPy also evaluates once
// per call (np.radians, focal length, cos/sin of the pitch: P:64-68, P:85, P:119, P:142-149).
#ifndef P2P_HOST_H
#define P2P_HOST_H
#include "../../include/p2p_hip.h"
#define P2P_HOST 1  // no tile-shape constants here: every shape through p2p::ShapeOps
#include "p2p_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

namespace p2p_host {

extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);  // sets p2p_last_error()'s text (no allocation), returns `code`

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? P2P_ERR_OOM : P2P_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                               \
    } while (0)

constexpr double kPi = 3.141592653589793;  // NPY_PI
inline double deg2rad(double d) { return d * (kPi / 180.0); }  // np.radians
constexpr size_t kSlack = 256;  // bytes past a panorama: the 8-byte pixel-pair loads may overrun by 5

// Every P2P_* environment knob of the library.  The environment is read ONCE per process, at the first entry point
// that needs a knob (and again only on reload_options(), which tests and tools call after changing a variable):
// getenv is not safe beside a host's setenv, and a launch parameter belongs to the job it was resolved for.  A job
// copies the per-job part at p2p_job_create; p2p_job_run, the memory pool and the kernels' dispatch read no
// environment.  -1 = "not set: the library's own rule applies".
struct Options {
    // process-wide
    long pool_mb = 8192;              // P2P_POOL_MB: idle bytes the device memory pool keeps per device
    int max_contexts = 64;            // P2P_MAX_CONTEXTS
    int oneshot_slots = 4;            // P2P_ONESHOT_SLOTS
    int oneshot_cache = 1;            // P2P_ONESHOT_CACHE
    long oneshot_cache_max_mb = 4096; // P2P_ONESHOT_CACHE_MAX_MB
    long plan_cache_mb = 4096;        // P2P_PLAN_CACHE_MB (a context copies it at p2p_ctx_create)
    // per job
    int plan_cache = 1;               // P2P_PLAN_CACHE
    int tile_shape = 0;               // P2P_TILE_SHAPE: 64 | 128 | 0 = choose_shape's rule
    int pairs_per_block = 0;          // P2P_PAIRS_PER_BLOCK
    int max_pairs_per_block = -1;     // P2P_MAX_PAIRS_PER_BLOCK
    int chunk_outer = -1;             // (no environment knob)
    int main_order = -1;              // P2P_MAIN_ORDER: 0 grid order, 1 list order, 2 list order also with several panoramas; -1 = by job
    int main_group = -1;              // P2P_MAIN_GROUP
    int main_tail = -1;               // P2P_MAIN_TAIL: list entries per XCD, at the end of its list, drawn by several workgroups each (-1: rule)
    int main_tail_parts = 2;          // P2P_MAIN_TAIL_PARTS: ... by how many (2..4)
    int main_span = -1;               // P2P_MAIN_SPAN: chunks of pairs one main-kernel workgroup draws (-1: the library's rule)
    int prefetch_lead = -1;           // P2P_PREFETCH_LEAD
    int force_rest = 0;               // (no environment knob)
    int gather_ppb = 16;              // P2P_GATHER_PPB
    int gather_all = 1;               // (no environment knob)
    int gather_blocky_from = p2p::GATHER_BLOCKY_FROM;  // P2P_GATHER_BLOCKY_FROM
    int gather_order = 1;             // P2P_GATHER_ORDER
    int gather_group = 3;             // (no environment knob)
    int scramble_plan = 0;            // P2P_SCRAMBLE_PLAN (robustness self-test only)
    int coords_all = 0;               // (no environment knob) 1 = the plan pass writes every pixel's quantised coordinates (0: the gather tiles')
    int merge_gather = 1;               // P2P_MERGE_GATHER: 1 = the gather tiles are drawn by the first workgroups of the band kernel's launch / of the main kernel's in list order
    int pair_ctx_table = 1;           // P2P_PAIR_CTX_TABLE: 1 = the pair contexts of every tile come from a table built once per job geometry
    int early_main = 1;               // P2P_EARLY_MAIN: 1 = a job's first launch sends the main kernel out right behind the plan pass
    int defer_lists = 2;              // P2P_DEFER_LISTS: where a fresh geometry's main lists (and the job's pair contexts) are made -- 0 in front of its first main kernel, 1 right behind it, 2 when a second launch asks
    int band = -1;                    // P2P_BAND: 1 = source-band tiles wherever they apply, 0 = never, -1 = the library's rule (choose_band)
    int band_bh = -1, band_cw = -1;   // P2P_BAND_BH / P2P_BAND_CW: cell of the source, rows x columns (-1: by tile shape, band_cell)
    int band_maxw = 27, band_maxh = 7;  // (no environment knob) tap extent of a group beyond which its tile gathers
};

Options options();  // a copy: callers keep what they resolved
void options_reload();

// the calling thread's current device, put back on scope exit: helpers that free another device's memory
// (pool trim, table destructors, p2p_release_cache) must not leave the caller on that device
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

// waits for a stream on every exit path that has not dismissed it: host vectors handed to hipMemcpyAsync and
// device blocks about to go back to the pool must not be in use by queued work when an error return unwinds
struct StreamSyncGuard {
    hipStream_t st;
    bool armed = true;
    explicit StreamSyncGuard(hipStream_t s) : st(s) {}
    ~StreamSyncGuard() { if (armed) (void)hipStreamSynchronize(st); }
};



// ---- device memory pool, pinned read-back blocks (p2p_host_pool.cpp) ----
hipError_t dev_alloc(void** out, size_t bytes) noexcept;  // a failure of the bookkeeping is hipErrorOutOfMemory
template <class T>
hipError_t dev_alloc(T** out, size_t bytes) noexcept { return dev_alloc((void**)out, bytes); }
hipError_t dev_free(void* ptr) noexcept;   // never throws: destructors call it
void dev_pool_trim();
void pool_set_budget(size_t bytes);
hipError_t pin_get(void** out, size_t* cls, size_t bytes);
void pin_put(void* p, size_t cls) noexcept;
void pin_pool_trim();
struct PinnedBlock {  // (declare BEFORE a StreamSyncGuard: the stream is drained before the block goes back)
    void* p = nullptr;
    size_t cls = 0;
    ~PinnedBlock() { pin_put(p, cls); }
};
int use_device(int device);
inline bool dims_ok(int w, int h) { return w >= 1 && h >= 1 && w < 32767 && h < 32767; }

// What the reference keeps for the life of the process in pitch_mapping_cache / yaw_mapping_cache (P:17-18, P:42-73):
// here the device tables built from a job geometry, kept by the CONTEXT and shared by every job on it that has
// the same key -- a second image of one geometry, on any path (one-shot, two-slot pipeline, view-sharded driver),
// launches nothing but the view kernels.
struct PlanKey {  // the reference's key (ow, oh, pitch, pw, ph, fov), for the whole pitch list, plus what shapes the tables
    int pw, ph, ow, oh, flags, border;
    double fov;
    std::vector<double> pitch;
    int shape;  // tile shape of the tables (0: 64 x 16, 1: 128 x 16)
    int knobs[11];  // the options that change the tables: gather_blocky_from (header bit), main_order (whether the main
                   // list exists), gather_order / gather_group (the XCD lists), band plan or not and its cell / extent
                   // parameters -- a plan built under one setting is never served under another
    bool operator<(const PlanKey& o) const
    {
        for (int i = 0; i < 11; ++i)
            if (knobs[i] != o.knobs[i]) return knobs[i] < o.knobs[i];
        if (shape != o.shape) return shape < o.shape;
        if (pw != o.pw) return pw < o.pw;
        if (ph != o.ph) return ph < o.ph;
        if (ow != o.ow) return ow < o.ow;
        if (oh != o.oh) return oh < o.oh;
        if (flags != o.flags) return flags < o.flags;
        if (border != o.border) return border < o.border;
        if (fov != o.fov) return fov < o.fov;
        return pitch < o.pitch;
    }
};

struct Plan {  // the plan pass's tables (p2p_plan.hip)
    int device = 0;
    void* d_block = nullptr;             // the one allocation the next six pointers are parts of
    int2* d_coords = nullptr;            // [n_pitch][oh][ow] quantised coordinates
    p2p::PieceHdr* d_hdr = nullptr;      // [n_pitch][tiles]
    uint32_t* d_px = nullptr;            // [n_pitch][tiles][256 * VIEWS_PXT]
    uint32_t* d_items = nullptr;         // [n_pitch][tiles][LDS_ITEMS_CAP]
    uint32_t* d_px2 = nullptr;           // float pixel path only: 16-bit coordinate fractions
    uint32_t* d_gather_list = nullptr;   // [n_pitch * tiles] tiles the plan marks for gathers (in the order the plan pass met them)
    uint32_t* d_xcd_list = nullptr;      // [8][xcd_stride] the same tiles dealt to the XCDs by source position (xcd_lists), ~0: none
    uint32_t* d_xcd_all = nullptr;       // [8][xcd_all_stride] every tile, likewise (only when most tiles gather: ViewsParams::gather_all)
    uint32_t* d_main_list = nullptr;     // [8][main_stride] the LDS-scheme tiles, dealt to the XCDs in source order (xcd_main_lists)
    uint32_t* d_main_count = nullptr;    // [8] entries of each XCD's main list (part of d_main_list's block; main_lists_kernel writes both)
    // what job_build_plan leaves behind without waiting for the stream: the host copies of the gather lists (their uploads
    // may still be queued) and the band passes' device scratch (their kernels may still be running); gone with the plan
    std::vector<uint32_t> h_xcd_list, h_xcd_all;
    std::vector<void*> build_blocks;
    int xcd_stride = 0, xcd_all_stride = 0, main_stride = 0;
    int n_gather = 0;
    // band plan (source-band tiles, p2p_device.h): no px / items tables; the band kernel draws band_tiles tiles
    bool band = false;
    p2p::PieceHdr* d_band_hdr = nullptr;
    uint32_t* d_band_px = nullptr;
    uint32_t* d_band_grp = nullptr;
    p2p::BandInfo* d_band_info = nullptr;
    int band_tiles = 0, band_groups = 0, band_per = 0;
    // The main kernel's per-XCD lists are made from the headers ON THE DEVICE (p2p_lists.hip: no read-back, no host sort,
    // no upload).  A plan with no gather tile does not need them to draw: its FIRST launch goes out in the grid's own
    // order right behind the plan pass, and the list kernel is enqueued when a second launch asks for the plan (one image
    // through a fresh context -- the tool on one file -- does not wait for it: bench.py's cold figures; lists_pending).
    // The device is busy from the plan pass to the end of the second launch either way; what P2P_DEFER_LISTS chooses is
    // which launch waits for the list kernel and the pair contexts (17 + 5 us).  One box, first / second launch of config 2:
    //   2 (the default)                                   174-177 / 104-107 us
    //   1: both right behind the first main kernel        196-207 /  81-84
    //   0: both between the plan pass and that kernel, which then draws in list order -- slower on a cold cache
    //                                                     202-207 /  81-89
    // the quantised coordinates of every pixel (else: of the gather tiles only; ensure_full_coords completes them)
    std::atomic<bool> coords_full{false};
    std::atomic<bool> lists_pending{false};
    bool lists_made = false;             // the main lists went out behind the plan pass, in front of the first launch (p2p_job_run)
    std::mutex lists_mu;
    std::atomic<int> launches{0};
    int tile_w = 64;
    bool built = false;
    float plan_ms = 0.0f;                // device time of the plan pass (band plans: with the band passes)
    size_t bytes = 0;
    unsigned long long stamp = 0;        // last use (eviction order)
    ~Plan()
    {
        DeviceRestore keep;
        (void)hipSetDevice(device);
        (void)dev_free(d_block);  // coords, hdr, px, items, px2, gather_list: parts of it
        (void)dev_free(d_xcd_list); (void)dev_free(d_xcd_all); (void)dev_free(d_main_list);
        (void)dev_free(d_band_hdr); (void)dev_free(d_band_px); (void)dev_free(d_band_grp); (void)dev_free(d_band_info);
        for (void* b : build_blocks)
            (void)dev_free(b);
    }
};


struct YawKey {  // the reference's key (pano_width, pano_height, yaw_angle), for the whole yaw list (rows do not depend on ph)
    int pw;
    std::vector<double> yaw;
    bool operator<(const YawKey& o) const { return pw != o.pw ? pw < o.pw : yaw < o.yaw; }
};

struct YawTabs {  // yaw_table_kernel / yaw_desc_kernel outputs
    int device = 0;
    uint32_t* d_ytab = nullptr;
    uint32_t* d_f4tab = nullptr;
    p2p::YawDesc* d_ydesc = nullptr;
    double* d_yaw_rad = nullptr;
    std::vector<p2p::YawDesc> desc;      // host copy (which yaws are odd)
    float tables_ms = 0.0f;
    size_t bytes = 0;
    unsigned long long stamp = 0;
    ~YawTabs()
    {
        DeviceRestore keep;
        (void)hipSetDevice(device);
        (void)dev_free(d_ytab); (void)dev_free(d_f4tab); (void)dev_free(d_ydesc); (void)dev_free(d_yaw_rad);
    }
};

}  // namespace p2p_host

struct p2p_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // kernels, and the synchronous copies
    // created on first use: a context that never copies asynchronously owns ONE hardware queue
    hipStream_t stream_up = nullptr;   // asynchronous panorama uploads (p2p_job_set_pano_async)
    hipStream_t stream_down = nullptr; // asynchronous view downloads (p2p_job_get_views_async)
    std::mutex stream_mu;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // p2p_ctx_mark
    short* d_ctab = nullptr;  // INTER_CUBIC weight table, built on first use
    // grow-only scratch of the generic remap entry point (source image, output, two maps): kept with the context
    // instead of four allocations per call
    void* scratch[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[4] = {0, 0, 0, 0};
    uint32_t* d_audit = nullptr;  // -DP2P_AUDIT builds: the kernels' violation record (p2p_audit.h)
    // geometry-keyed table caches (see p2p_host::PlanKey / p2p_host::YawKey); entries no job refers to go first when the byte budget
    // (P2P_PLAN_CACHE_MB, default 4096) is exceeded
    std::mutex cache_mu;
    std::map<p2p_host::PlanKey, std::shared_ptr<p2p_host::Plan>> plans;
    std::map<p2p_host::YawKey, std::shared_ptr<p2p_host::YawTabs>> yaw_tabs;
    unsigned long long cache_clock = 0;
    size_t cache_budget = (size_t)4096 << 20;     // P2P_PLAN_CACHE_MB as it stood when the context was created
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // timing of the plan pass / the table kernels
    // per-view plan passes: the gather counter [PLAN_TICKET_WORDS] and the finish counters [0 .. PLAN_TICKET_WORDS) of
    // plan_kernel, zero whenever no pass is in flight (the pass's last workgroup leaves them so); one pass at a time
    uint32_t* d_plan_cnt = nullptr;
    std::mutex plan_mu;
};



struct p2p_job {
    p2p_ctx* ctx = nullptr;
    p2p_job_desc d{};              // sizes and flags (the angle pointers are not kept: see yaw / pitch / fov)
    std::vector<double> yaw, pitch;  // degrees, any real value (P:85 and P:64-68 go through np.radians)
    double fov = 90.0;
    uint8_t* d_src = nullptr;
    bool owns_src = true;            // false: the panoramas are another job's (p2p_job_share_panos)
    p2p_job* src_owner = nullptr;
    // ordering between the context's three streams (all created with hipEventDisableTiming):
    hipEvent_t ev_up = nullptr;      // last asynchronous upload into d_src     -> the next run waits for it
    hipEvent_t ev_run = nullptr;     // last run                                -> uploads and downloads wait for it
    hipEvent_t ev_down = nullptr;    // last asynchronous download from d_out   -> the next run waits for it
    bool up_pending = false, down_pending = false;
    bool ev_run_recorded = false;    // ev_run has been recorded at least once
    bool run_unmarked = false;       // a run (of this job or of one that borrows its panoramas) was enqueued after it
    size_t pano_stride = 0;
    int src_pitch = 0;
    uint8_t* d_out = nullptr;
    size_t out_bytes = 0;            // device bytes of all views: n_views * oh * out_row
    uint8_t* d_pack = nullptr;       // odd view widths only: one panorama's views without the row padding, for the download            // device bytes of all views: n_views * oh * out_row
    int out_row = 0;                 // bytes per view row on the device (12-byte groups: ViewsParams::out_row)
    std::shared_ptr<p2p_host::Plan> plan_ref;      // owns the plan tables below (shared through the context's cache, or private)
    std::shared_ptr<p2p_host::YawTabs> yaw_ref;    // owns the yaw tables below
    uint32_t* d_ytab = nullptr;
    uint32_t* d_f4tab = nullptr;
    p2p::YawDesc* d_ydesc = nullptr;
    double* d_yaw_rad = nullptr;
    p2p::PitchConst* d_pitch = nullptr;
    float* d_mapU = nullptr;
    float* d_mapV = nullptr;
    float* d_rows = nullptr;
    // the plan (p2p_plan.hip): what depends on the maps only, built once per job geometry like the reference's
    // pitch_mapping_cache (P:17-18, P:55-73)
    int2* d_coords = nullptr;            // [n_pitch][oh][ow] quantised coordinates
    p2p::PieceHdr* d_hdr = nullptr;      // [n_pitch][tiles]
    uint32_t* d_px = nullptr;            // [n_pitch][tiles][256 * VIEWS_PXT]
    uint32_t* d_items = nullptr;         // [n_pitch][tiles][LDS_ITEMS_CAP]
    uint32_t* d_px2 = nullptr;           // float pixel path only: 16-bit coordinate fractions
    uint32_t* d_gather_list = nullptr;   // [n_pitch * tiles] tiles the plan marks for gathers
    uint32_t* d_xcd_list = nullptr, *d_xcd_all = nullptr;  // the gather kernel's per-XCD work lists (see p2p_host::Plan)
    int xcd_stride = 0, xcd_all_stride = 0;
    uint32_t* d_main_list = nullptr;     // the main kernel's per-XCD work lists (see p2p_host::Plan)
    int main_stride = 0;
    const uint32_t* d_main_count = nullptr;
    uint32_t* d_odd_pairs = nullptr;     // (panorama, yaw) pairs whose yaw is not a plain shift with one weight
    int n_odd_pairs = 0;
    int n_gather = 0;                    // tiles the plan marks for gathers
    int row0 = 0, row1 = 0;              // output rows the job draws, [row0, row1) of every view (p2p_job_set_rows; row1 = oh at creation)
    int n_odd_yaws = 0;                  // yaws that are not a plain shift with one weight (YawDesc.mode != 0)
    uint16_t* d_pitch_order = nullptr;   // [n_pitch] heaviest view first
    // the pair-context table (p2p_views.hip: pair_ctx_kernel): what every workgroup would work out about its chunk's
    // (panorama, yaw) pairs, once per (plan, yaw tables, view mask, pairs per workgroup) -- rebuilt when one of them changes
    uint4* d_pair_ctx = nullptr;
    const void* pc_plan = nullptr;
    const void* pc_yaw = nullptr;
    unsigned long long pc_mask_gen = 0, mask_gen = 1;
    int pc_ppb = 0, pc_chunks = 0;
    size_t pc_slots = 0;
    uint32_t* d_view_mask = nullptr;     // sparse view sets (p2p_job_set_view_mask): [n_pitch][mask_words] bits, nullptr = every view
    int mask_words = 0;
    int n_views_wanted = 0;              // views per panorama the job draws (n_yaw * n_pitch without a mask)
    size_t n_tiles = 0;
    int shape = 0;                       // tile shape of the job's plan and kernels (choose_shape)
    p2p::MapGeom geom{};
    bool host_maps = false;
    unsigned long long maps_key = 0;  // caller's name for the maps the job holds (p2p_remap_views_pitch_maps_f64); 0: none
    bool rows_from_host = false;  // yaw tables were packed from caller float rows, not built from yaw_deg
    bool time_launches = false; // bracket every launch with its own event pair (p2p_job_time_launches / p2p_job_kernel_ms*)
    p2p_host::Options opt;                // the knobs as they stood at job_create (no environment is read after that)
    int border = 0;             // stage-2 border mode; non-zero only for the legacy single-remap entry point
    bool ran = false;
    std::vector<char> pano_set;
    // ring of event pairs, one per p2p_job_run, so that a caller can time K back-to-back launches without
    // synchronising between them (bench.py's roofline figure).  It exists only while job_time_launches(job, n)
    // has asked for n pairs: a job that nobody times creates no timing event and records none.
    std::vector<hipEvent_t> ev_ring;  // 2 * ring_pairs events
    int ring_pairs = 0;
    long long runs = 0;
};


namespace p2p_host {

static constexpr int kEvRingMax = 4096;

// ---- contexts, caches, yaw tables (p2p_host_ctx.cpp) ----
hipError_t ctx_copy_stream(p2p_ctx* c, bool up, hipStream_t* out);
size_t cache_evict_unused(p2p_ctx* c, size_t budget);
size_t caches_evict_all();
void cache_trim(p2p_ctx* c);  // (cache_mu held by the caller)
int job_adopt_yaw_tabs(p2p_job* j, std::shared_ptr<YawTabs> yt);
int yaw_tabs_get(p2p_ctx* ctx, int pw, const std::vector<double>& yaw_deg, const float* rows, float* d_rows,
                 bool use_cache, std::shared_ptr<YawTabs>* out, const std::function<void()>& while_device_works = nullptr);

// ---- shapes, chunking, plans, work lists (p2p_host_plan.cpp) ----
const p2p::ShapeOps& shape_ops(int shape);
void band_cell(const Options& o, int shape, int* bh, int* cw);
int choose_shape(const p2p_job_desc& d, const Options& opt);
int choose_pairs_per_block(const p2p_job_desc& d, const p2p::TileShape& S, const Options& opt);
size_t plan_table_bytes(const p2p_job_desc& d, const p2p::TileShape& S);
int choose_main_span(const p2p_job_desc& d, const p2p::TileShape& S, const Options& opt, int pairs_per_block);
int choose_main_group(const Options& opt, int shape, int span, int chunks);
bool job_wants_band(const p2p_job* j);
void job_settle_shape(p2p_job* j);
int job_main_order(const p2p_job* j);
int job_build_plan(p2p_job* j, const std::function<int(Plan&)>& after_plan_pass = nullptr);
int ensure_full_coords(p2p_job* j);
int plan_make_main_lists(p2p_job* j, Plan& Pl);
void plan_block_prefetch(p2p_job* j) noexcept;
int plan_enqueue_main_lists(Plan& Pl, size_t slots, int tile_w, hipStream_t st);

// ---- implementations of the ABI functions: p2p_xyz -> p2p_host::xyz ----
const char* version();
const char* last_error();
int device_count();
int remap_views_u8(const uint8_t* pano, int pw, int ph, int64_t row_stride, const int32_t* yaw_deg, int n_yaw,
                   const int32_t* pitch_deg, int n_pitch, int fov_deg, int ow, int oh, uint8_t* out, int device,
                   int flags);
int remap_views_f64(const uint8_t* pano, int pw, int ph, int64_t row_stride, const double* yaw_deg, int n_yaw,
                    const double* pitch_deg, int n_pitch, double fov_deg, int ow, int oh, uint8_t* out, int device,
                    int flags);
int remap_views_maps_u8(const uint8_t* pano, int pw, int ph, int64_t row_stride, const float* yaw_rows, int n_yaw,
                        const float* U, const float* V, int n_pitch, int ow, int oh, uint8_t* out, int device);
int remap_views_pitch_maps_f64(const uint8_t* pano, int pw, int ph, int64_t row_stride, const double* yaw_deg, int n_yaw,
                               const float* U, const float* V, int n_pitch, uint64_t maps_key, int ow, int oh,
                               uint8_t* out, int device);
int remap_maps_u8(const uint8_t* src, int sw, int sh, int64_t row_stride, int cn, const float* U, const float* V, int ow,
                  int oh, uint8_t* out, int border_mode, const uint8_t* border_value, int device);
int remap_maps_batch_u8(const uint8_t* src, int sw, int sh, int64_t row_stride, const float* U, const float* V,
                        int n_maps, int ow, int oh, uint8_t* out, int border_mode, int device);
int remap_maps_interp_u8(const uint8_t* src, int sw, int sh, int64_t row_stride, int cn, const float* U, const float* V,
                         int ow, int oh, uint8_t* out, int interpolation, int border_mode, const uint8_t* border_value,
                         int device);
int build_pitch_map(int ow, int oh, double fov_rad, double pitch_rad, int pw, int ph, float* U, float* V, int device);
int build_rot_map(int ow, int oh, double fov_rad, const float* R9, int pw, int ph, float* U, float* V, int device);
int build_yaw_row(int pw, double yaw_rad, float* U_row, int device);
int ctx_create(int device, p2p_ctx** out);
void ctx_destroy(p2p_ctx* ctx);
int ctx_synchronize(p2p_ctx* ctx);
int ctx_mark(p2p_ctx* ctx, int which);
int ctx_marked_ms(p2p_ctx* ctx, float* ms);
int job_create(p2p_ctx* ctx, const p2p_job_desc* desc, p2p_job** out);
int job_create_f64(p2p_ctx* ctx, const p2p_job_desc_f64* desc, p2p_job** out);
void job_destroy(p2p_job* job);
int job_set_pano(p2p_job* job, int index, const uint8_t* pano, int64_t row_stride);
int job_set_pano_async(p2p_job* job, int index, const uint8_t* pano, int64_t row_stride);
int job_share_panos(p2p_job* job, p2p_job* owner);
int job_set_yaws(p2p_job* job, const int32_t* yaw_deg);
int job_set_yaws_f64(p2p_job* job, const double* yaw_deg);
int job_set_maps(p2p_job* job, const float* yaw_rows, const float* U, const float* V);
int job_set_border(p2p_job* job, int border_mode);
int job_set_view_mask(p2p_job* job, const uint8_t* mask);
int job_run(p2p_job* job);
int job_get_views(p2p_job* job, int index, uint8_t* out);
int job_get_views_async(p2p_job* job, int index, uint8_t* out);
int job_get_view(p2p_job* job, int index, int yaw_i, int pitch_i, uint8_t* out);
int job_get_view_async(p2p_job* job, int index, int yaw_i, int pitch_i, uint8_t* out);
int job_set_rows(p2p_job* job, int row0, int row1);
int job_get_view_rows(p2p_job* job, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out);
int job_get_view_rows_async(p2p_job* job, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out);
int job_wait(p2p_job* job);
int job_time_launches(p2p_job* job, int n);
int job_plan_ms(p2p_job* job, float* plan_ms, float* tables_ms);
int job_kernel_ms(p2p_job* job, float* ms);
int job_kernel_ms_last(p2p_job* job, float* ms, int n);
void* job_device_out(p2p_job* job, int64_t* bytes);
int job_get_coords(p2p_job* job, int32_t* sxsy);
int job_get_yaw_tables(p2p_job* job, uint32_t* packed);
int job_get_info(p2p_job* job, p2p_job_info* out);
int host_alloc(size_t bytes, void** out);
int host_free(void* ptr);
int release_cache();
int reload_options();
int device_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes);
}  // namespace p2p_host
#endif  // P2P_HOST_H
