// p2p_host_pool.cpp -- error text, the P2P_* options, the device memory pool, the pinned read-back blocks.
// Part of the host side of libp2p_hip.so (see p2p_host.h for the units); C ABI: include/p2p_hip.h via p2p_abi.cpp.
#include "p2p_host.h"

namespace p2p_host {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

std::mutex g_opt_mu;
Options g_opt;
bool g_opt_loaded = false;

void options_load_locked()
{
    Options o;
    o.pool_mb = std::max(0, env_int("P2P_POOL_MB", (int)o.pool_mb));
    o.max_contexts = std::max(1, env_int("P2P_MAX_CONTEXTS", o.max_contexts));
    o.oneshot_slots = std::max(1, env_int("P2P_ONESHOT_SLOTS", o.oneshot_slots));
    o.oneshot_cache = env_int("P2P_ONESHOT_CACHE", o.oneshot_cache);
    o.oneshot_cache_max_mb = std::max(0, env_int("P2P_ONESHOT_CACHE_MAX_MB", (int)o.oneshot_cache_max_mb));
    o.plan_cache_mb = std::max(0, env_int("P2P_PLAN_CACHE_MB", (int)o.plan_cache_mb));
    o.plan_cache = env_int("P2P_PLAN_CACHE", o.plan_cache);
    o.tile_shape = env_int("P2P_TILE_SHAPE", o.tile_shape);
    o.pairs_per_block = env_int("P2P_PAIRS_PER_BLOCK", o.pairs_per_block);
    o.max_pairs_per_block = env_int("P2P_MAX_PAIRS_PER_BLOCK", o.max_pairs_per_block);
    o.main_order = env_int("P2P_MAIN_ORDER", o.main_order);
    o.main_group = env_int("P2P_MAIN_GROUP", o.main_group);
    o.main_span = env_int("P2P_MAIN_SPAN", o.main_span);
    o.main_tail = env_int("P2P_MAIN_TAIL", o.main_tail);
    o.main_tail_parts = std::min(4, std::max(2, env_int("P2P_MAIN_TAIL_PARTS", o.main_tail_parts)));
    o.prefetch_lead = env_int("P2P_PREFETCH_LEAD", o.prefetch_lead);
    o.gather_ppb = env_int("P2P_GATHER_PPB", o.gather_ppb);
    o.gather_blocky_from = std::max(0, env_int("P2P_GATHER_BLOCKY_FROM", o.gather_blocky_from));
    o.gather_order = env_int("P2P_GATHER_ORDER", o.gather_order);
    o.scramble_plan = env_int("P2P_SCRAMBLE_PLAN", o.scramble_plan);
    o.defer_lists = env_int("P2P_DEFER_LISTS", o.defer_lists);
    o.early_main = env_int("P2P_EARLY_MAIN", o.early_main);
    o.pair_ctx_table = env_int("P2P_PAIR_CTX_TABLE", o.pair_ctx_table);
    o.merge_gather = env_int("P2P_MERGE_GATHER", o.merge_gather);
    o.band = env_int("P2P_BAND", o.band);
    o.band_bh = std::min(256, env_int("P2P_BAND_BH", o.band_bh));
    o.band_cw = std::min(256, env_int("P2P_BAND_CW", o.band_cw));
    g_opt = o;
    g_opt_loaded = true;
    pool_set_budget((size_t)o.pool_mb << 20);
}

void options_reload()
{
    std::lock_guard<std::mutex> lk(g_opt_mu);
    options_load_locked();
}

Options options()  // a copy: callers keep what they resolved
{
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (!g_opt_loaded)
        options_load_locked();
    return g_opt;
}

// ------------------------------------------------------------------------------------------
// Device memory pool.  Every device buffer of the library comes from here and goes back here: nothing returns to
// the driver in steady state.  Not (only) a speed matter: on the round-3 GPU pool, device memory that has JUST been
// allocated loses what a kernel or a host copy wrote into it -- page-sized runs read back as zeros or as other
// data -- about once in 6 000 allocate-write-check-free rounds when several processes allocate and free on one GPU
// at the same time (tools/platform/alloc_churn.hip reproduces it with no code of this library: 345 bad rounds in
// 2.1 M; a buffer allocated once and reused: 0 in 0.9 M beside the same neighbours).  That is what round 2's
// "wrong output, then a memory fault" was: a plan table with garbage in it.  The kernels now clamp or range-check
// every table-derived offset (p2p_audit.h), and with the pool a fresh allocation happens only while the pool warms up.
// Blocks are kept in size classes (<= 12.5 % rounding), P2P_POOL_MB (default 8192) bounds the idle bytes per
// device (the largest idle blocks go back to the driver first), p2p_release_cache empties it, and so does the
// destruction of the process's last context.
// A block goes back to the idle list only when nothing queued on the device can still touch it: every caller either
// has synchronised the streams that used it (job / context destruction, table replacement) or frees behind a
// StreamSyncGuard on its error paths.
// ------------------------------------------------------------------------------------------

struct DevPool {
    std::mutex mu;
    std::atomic<size_t> budget{(size_t)8192 << 20};  // idle bytes kept per device (Options::pool_mb, set when the options load)
    struct PerDev {
        std::multimap<size_t, void*> idle;        // class size -> block
        std::map<void*, size_t> live;             // block -> class size
        size_t idle_bytes = 0;
    };
    std::map<int, PerDev> dev;
};

DevPool& dev_pool()
{
    static DevPool* p = new DevPool();  // never destroyed: no HIP call after the runtime's own shutdown
    return *p;
}

size_t pool_class(size_t bytes)
{
    if (bytes < 256)
        return 256;
    int lg = 63 - __builtin_clzll((unsigned long long)bytes);
    if ((size_t)1 << lg == bytes)
        return bytes;
    const size_t step = lg >= 20 ? (size_t)1 << (lg - 3) : (size_t)1 << lg;  // eighths of a power of two from 1 MB up
    return (bytes + step - 1) / step * step;
}


// Pinned host blocks for the plan's read-backs (counters, band counts, tile headers): a hipMemcpyAsync into pageable memory
// is staged and waited for one copy at a time -- three of them were 60 us of a cold band plan.  Process-wide, power-of-two
// classes from 64 KB; hipHostMalloc itself costs hundreds of microseconds, so blocks come back here (p2p_release_cache
// frees the idle ones).
struct PinPool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> idle;
};
PinPool& pin_pool() { static PinPool* P = new PinPool; return *P; }
hipError_t pin_get(void** out, size_t* cls, size_t bytes)
{
    size_t c = (size_t)64 << 10;
    while (c < bytes)
        c <<= 1;
    *cls = c;
    PinPool& P = pin_pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (size_t i = 0; i < P.idle.size(); ++i)
            if (P.idle[i].second == c) {
                *out = P.idle[i].first;
                P.idle.erase(P.idle.begin() + (long)i);
                return hipSuccess;
            }
    }
    // (portable and mapped: the plan pass of ANY device writes its counts and headers into these blocks through the block's own
    // address -- job_build_plan)
    return hipHostMalloc(out, c, hipHostMallocPortable | hipHostMallocMapped);
}
void pin_put(void* p, size_t cls) noexcept
{
    if (!p)
        return;
    PinPool& P = pin_pool();
    try {
        std::lock_guard<std::mutex> lk(P.mu);
        if (P.idle.size() < 16) {
            P.idle.emplace_back(p, cls);
            return;
        }
    } catch (...) {  // (no room for the bookkeeping: the block goes back to the runtime instead)
    }
    (void)hipHostFree(p);
}
void pin_pool_trim()
{
    PinPool& P = pin_pool();
    std::vector<std::pair<void*, size_t>> drop;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        drop.swap(P.idle);
    }
    for (auto& b : drop)
        (void)hipHostFree(b.first);
}
// Never throws: a host allocation failure inside the pool's bookkeeping comes back as hipErrorOutOfMemory, and the block
// it was about goes back where it came from.
hipError_t dev_alloc(void** out, size_t bytes) noexcept
{
    *out = nullptr;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess)
        return e;
    const size_t cls = pool_class(bytes);
    DevPool& P = dev_pool();
    try {
        std::lock_guard<std::mutex> lk(P.mu);
        DevPool::PerDev& D = P.dev[device];
        auto it = D.idle.find(cls);
        if (it != D.idle.end()) {
            D.live[it->second] = cls;  // (first: if the entry cannot be made the block simply stays idle)
            *out = it->second;
            D.idle.erase(it);
            D.idle_bytes -= cls;
            return hipSuccess;
        }
    } catch (...) {
        return hipErrorOutOfMemory;
    }
    e = hipMalloc(out, cls);
    if (e != hipSuccess) {
        // out of device memory: the cached tables no job uses go to the pool, the pool's idle blocks go back to the
        // driver, and the allocation is tried once more
        (void)hipGetLastError();
        try {
            (void)caches_evict_all();
        } catch (...) {
        }
        for (;;) {  // (block by block: no list of them to allocate)
            void* q = nullptr;
            {
                std::lock_guard<std::mutex> lk(P.mu);
                auto dv = P.dev.find(device);
                if (dv == P.dev.end() || dv->second.idle.empty())
                    break;
                auto it = dv->second.idle.begin();
                q = it->second;
                dv->second.idle_bytes -= it->first;
                dv->second.idle.erase(it);
            }
            (void)hipFree(q);
        }
        (void)hipGetLastError();
        e = hipMalloc(out, cls);
        if (e != hipSuccess) {
            *out = nullptr;
            return e;
        }
    }
    try {
        std::lock_guard<std::mutex> lk(P.mu);
        P.dev[device].live[*out] = cls;
    } catch (...) {
        (void)hipFree(*out);
        *out = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipSuccess;
}

// Destructors call this (Plan, YawTabs, a job's teardown): it never throws.  If the idle list cannot take the block -- a
// host allocation failure inside the bookkeeping -- the block goes back to the driver instead of to the pool.
hipError_t dev_free(void* ptr) noexcept
{
    if (!ptr)
        return hipSuccess;
    DevPool& P = dev_pool();
    void* drop[8];
    int n_drop = 0;
    bool ours = false, kept = false;
    try {
        std::lock_guard<std::mutex> lk(P.mu);
        for (auto& dv : P.dev) {
            auto it = dv.second.live.find(ptr);
            if (it == dv.second.live.end())
                continue;
            const size_t cls = it->second;
            dv.second.live.erase(it);
            ours = true;
            dv.second.idle.emplace(cls, ptr);  // (may throw: caught below, the block is then freed outright)
            kept = true;
            dv.second.idle_bytes += cls;
            const size_t budget = P.budget.load(std::memory_order_relaxed);
            while (dv.second.idle_bytes > budget && !dv.second.idle.empty() && n_drop < 8) {  // largest idle blocks first
                auto big = std::prev(dv.second.idle.end());
                dv.second.idle_bytes -= big->first;
                drop[n_drop++] = big->second;
                dv.second.idle.erase(big);
            }
            break;
        }
    } catch (...) {
    }
    for (int i = 0; i < n_drop; ++i) (void)hipFree(drop[i]);
    if (!kept)
        return hipFree(ptr);  // the bookkeeping failed, or not ours (cannot happen through this library)
    (void)ours;
    return hipSuccess;
}

void pool_set_budget(size_t bytes) { dev_pool().budget.store(bytes, std::memory_order_relaxed); }

void dev_pool_trim()
{
    DevPool& P = dev_pool();
    std::vector<std::pair<int, void*>> drop;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (auto& dv : P.dev) {
            for (auto& kv : dv.second.idle) drop.emplace_back(dv.first, kv.second);
            dv.second.idle.clear();
            dv.second.idle_bytes = 0;
        }
    }
    DeviceRestore keep;
    for (auto& d : drop) {
        (void)hipSetDevice(d.first);
        (void)hipFree(d.second);
    }
}



int use_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(P2P_ERR_NO_DEVICE, "no HIP device is available (hipGetDeviceCount found none)");
    if (device < 0 || device >= n)
        return fail(P2P_ERR_NO_DEVICE, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    return P2P_OK;
}

}  // namespace p2p_host
