// p2p_host_ctx.cpp -- contexts (one device, one stream, four events), their geometry-keyed table caches (the reference's
// yaw_mapping_cache / pitch_mapping_cache, P:17-18, P:42-73, on the device) and the yaw tables.
// Part of the host side of libp2p_hip.so (see p2p_host.h for the units); C ABI: include/p2p_hip.h via p2p_abi.cpp.
#include "p2p_host.h"

namespace p2p_host {

// every live context, so that p2p_release_cache and an out-of-memory retry can reach the table caches of contexts
// the caller created itself (lock order: registry, then a context's cache_mu)
struct CtxRegistry {
    std::mutex mu;
    std::vector<p2p_ctx*> all;
};
CtxRegistry& ctx_registry()
{
    static CtxRegistry* r = new CtxRegistry();
    return *r;
}

// the context's copy streams, created when an asynchronous copy first asks for one
hipError_t ctx_copy_stream(p2p_ctx* c, bool up, hipStream_t* out)
{
    std::lock_guard<std::mutex> lk(c->stream_mu);
    hipStream_t& st = up ? c->stream_up : c->stream_down;
    if (!st) {
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) {
            st = nullptr;
            return e;
        }
    }
    *out = st;
    return hipSuccess;
}

// drop the cached tables of `c` that no job refers to (all of them, or until `budget` holds); returns bytes dropped.
// The tables' device blocks go back to the pool: the caller trims the pool if the driver should have them.
size_t cache_evict_unused(p2p_ctx* c, size_t budget);



const char* version(void) { return "0.3.0-gfx950"; }
const char* last_error(void) { return g_err; }

int device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n < 0 ? 0 : n;
}

std::atomic<int>& live_contexts()
{
    static std::atomic<int> n{0};
    return n;
}

int ctx_create(int device, p2p_ctx** out)
{
    if (!out)
        return fail(P2P_ERR_INVALID, "p2p_ctx_create: out is NULL");
    *out = nullptr;
    int rc = use_device(device);
    if (rc != P2P_OK)
        return rc;
    // A context owns a HIP stream (a hardware queue; two more once it copies asynchronously) and four events.  A
    // process that creates them without bound takes the GPU down for everybody (round 3: a test script with nine
    // thousand threads, a context each): beyond P2P_MAX_CONTEXTS live contexts the call fails instead.  The default,
    // 64, is what has run on this pool without incident (48 threads x 4 one-shot slots, 8 explicit contexts per test);
    // nothing larger has been tried on a GPU, so nothing larger is the default.
    const Options opt = options();
    const int max_ctx = opt.max_contexts;
    if (live_contexts().fetch_add(1) >= max_ctx) {
        live_contexts().fetch_sub(1);
        return fail(P2P_ERR_OOM, "p2p_ctx_create: %d contexts are alive in this process (P2P_MAX_CONTEXTS)", max_ctx);
    }
    p2p_ctx* c = new (std::nothrow) p2p_ctx();
    if (!c) {
        live_contexts().fetch_sub(1);
        return fail(P2P_ERR_OOM, "host allocation failed");
    }
    c->device = device;
    c->cache_budget = (size_t)opt.plan_cache_mb << 20;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_t0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_t1);
    if (e == hipSuccess) e = dev_alloc((void**)&c->d_plan_cnt, (p2p::PLAN_TICKET_WORDS + 32) * sizeof(uint32_t));
    if (e == hipSuccess) e = p2p::launch_zero_words(c->d_plan_cnt, p2p::PLAN_TICKET_WORDS + 32, c->stream);
#ifdef P2P_AUDIT
    if (e == hipSuccess) e = dev_alloc((void**)&c->d_audit, p2p::AUDIT_WORDS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(c->d_audit, 0, p2p::AUDIT_WORDS * sizeof(uint32_t));
#endif
    if (e != hipSuccess) {
        ctx_destroy(c);
        return fail(P2P_ERR_HIP, "stream/event creation: %s", hipGetErrorString(e));
    }
    try {
        CtxRegistry& R = ctx_registry();
        std::lock_guard<std::mutex> lk(R.mu);
        R.all.push_back(c);
    } catch (...) {
        ctx_destroy(c);  // (not registered: the erase in there finds nothing)
        throw;
    }
    *out = c;
    return P2P_OK;
}

void ctx_destroy(p2p_ctx* c)
{
    if (!c)
        return;
    {
        CtxRegistry& R = ctx_registry();
        std::lock_guard<std::mutex> lk(R.mu);
        R.all.erase(std::remove(R.all.begin(), R.all.end(), c), R.all.end());
    }
    const bool last = live_contexts().fetch_sub(1) == 1;
    DeviceRestore keep;
    (void)hipSetDevice(c->device);
    for (hipStream_t* st : {&c->stream_up, &c->stream_down, &c->stream})
        if (*st) {
            (void)hipStreamSynchronize(*st);
            (void)hipStreamDestroy(*st);
        }
    {
        std::lock_guard<std::mutex> lk(c->cache_mu);
        c->plans.clear();     // (jobs still alive keep their tables through their own references)
        c->yaw_tabs.clear();
    }
    (void)dev_free(c->d_ctab);
    (void)dev_free(c->d_plan_cnt);
    (void)dev_free(c->d_audit);
    for (void* p : c->scratch)
        (void)dev_free(p);
    for (hipEvent_t e : {c->ev0, c->ev1, c->ev_t0, c->ev_t1})
        if (e) (void)hipEventDestroy(e);
    delete c;
    if (last)
        dev_pool_trim();  // the process's last context: the idle blocks go back to the driver (co-tenants, torch)
}

int ctx_synchronize(p2p_ctx* c)
{
    if (!c)
        return fail(P2P_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (c->stream_up) HIP_TRY(hipStreamSynchronize(c->stream_up));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->stream_down) HIP_TRY(hipStreamSynchronize(c->stream_down));
    return P2P_OK;
}


// ---- the context's table caches ------------------------------------------------------------------------------
// drop cached tables nobody uses, least recently used first, until `budget` holds (mutex held by the caller);
// returns the bytes dropped
size_t cache_trim_to(p2p_ctx* c, size_t budget)
{
    size_t total = 0, dropped = 0;
    for (auto& kv : c->plans) total += kv.second->bytes;
    for (auto& kv : c->yaw_tabs) total += kv.second->bytes;
    while (total > budget) {
        unsigned long long best = ~0ull;
        int which = 0;
        std::map<PlanKey, std::shared_ptr<Plan>>::iterator bp;
        std::map<YawKey, std::shared_ptr<YawTabs>>::iterator by;
        for (auto it = c->plans.begin(); it != c->plans.end(); ++it)
            if (it->second.use_count() == 1 && it->second->stamp < best) { best = it->second->stamp; bp = it; which = 1; }
        for (auto it = c->yaw_tabs.begin(); it != c->yaw_tabs.end(); ++it)
            if (it->second.use_count() == 1 && it->second->stamp < best) { best = it->second->stamp; by = it; which = 2; }
        if (!which)
            break;  // everything left is in use
        if (which == 1) { total -= bp->second->bytes; dropped += bp->second->bytes; c->plans.erase(bp); }
        else { total -= by->second->bytes; dropped += by->second->bytes; c->yaw_tabs.erase(by); }
    }
    return dropped;
}
void cache_trim(p2p_ctx* c) { (void)cache_trim_to(c, c->cache_budget); }

size_t cache_evict_unused(p2p_ctx* c, size_t budget)
{
    // an entry nobody uses was last touched by a job that has synchronised the context's streams since (job
    // destruction, table replacement): nothing queued reads it
    std::unique_lock<std::mutex> lk(c->cache_mu, std::try_to_lock);
    if (!lk.owns_lock())
        return 0;  // its owner is inserting right now: leave it
    return cache_trim_to(c, budget);
}

// every live context's unused cached tables -> the pool (p2p_release_cache; dev_alloc's out-of-memory retry)
size_t caches_evict_all()
{
    size_t dropped = 0;
    CtxRegistry& R = ctx_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    for (p2p_ctx* c : R.all)
        dropped += cache_evict_unused(c, 0);
    return dropped;
}

// point the job's table pointers at its (shared or private) YawTabs and list its odd pairs: every panorama x the
// yaws with per-column weights or rows that are not a shift
int job_adopt_yaw_tabs(p2p_job* j, std::shared_ptr<YawTabs> yt)
{
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));  // nothing in flight reads the old tables or the old list
    j->yaw_ref = std::move(yt);
    const YawTabs& T = *j->yaw_ref;
    j->d_ytab = T.d_ytab; j->d_f4tab = T.d_f4tab; j->d_ydesc = T.d_ydesc; j->d_yaw_rad = T.d_yaw_rad;
    j->n_odd_yaws = 0;
    for (const auto& d : T.desc)
        j->n_odd_yaws += d.mode != 0;
    (void)dev_free(j->d_odd_pairs);
    j->d_odd_pairs = nullptr;
    j->n_odd_pairs = 0;
    if (j->n_odd_yaws > 0) {
        std::vector<uint32_t> pairs;
        for (int p = 0; p < j->d.n_panos; ++p)
            for (int y = 0; y < j->d.n_yaw; ++y)
                if (T.desc[y].mode != 0)
                    pairs.push_back((uint32_t)p * (uint32_t)j->d.n_yaw + (uint32_t)y);
        HIP_TRY(dev_alloc((void**)&j->d_odd_pairs, pairs.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(j->d_odd_pairs, pairs.data(), pairs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        j->n_odd_pairs = (int)pairs.size();
    }
    return P2P_OK;
}

// Yaw tables for a list of yaw angles (degrees) or for caller float rows: built by yaw_table_kernel /
// yaw_pack_kernel + yaw_desc_kernel on the context's stream.  rows == nullptr: looked up in / entered into the
// context's cache (the reference's yaw_mapping_cache, P:17, P:42-52); caller rows make private tables.
// while_device_works: called once the table kernels are queued and before they are waited for (host work that may as
// well happen meanwhile: p2p_job_create fetches the block of the job's plan); not called when the tables were cached.
int yaw_tabs_get(p2p_ctx* ctx, int pw, const std::vector<double>& yaw_deg, const float* rows, float* d_rows,
                        bool use_cache, std::shared_ptr<YawTabs>* out, const std::function<void()>& while_device_works)
{
    const int n_yaw = (int)yaw_deg.size();
    YawKey key{pw, yaw_deg};
    const bool cached = rows == nullptr && use_cache;
    if (cached) {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        auto it = ctx->yaw_tabs.find(key);
        if (it != ctx->yaw_tabs.end()) {
            it->second->stamp = ++ctx->cache_clock;
            *out = it->second;
            return P2P_OK;
        }
    }
    auto T = std::make_shared<YawTabs>();
    T->device = ctx->device;
    const size_t n = (size_t)n_yaw * pw;
    HIP_TRY(dev_alloc((void**)&T->d_ytab, n * sizeof(uint32_t)));
    HIP_TRY(dev_alloc((void**)&T->d_f4tab, n * sizeof(uint32_t)));
    HIP_TRY(dev_alloc((void**)&T->d_ydesc, (size_t)n_yaw * sizeof(p2p::YawDesc)));
    HIP_TRY(dev_alloc((void**)&T->d_yaw_rad, (size_t)n_yaw * sizeof(double)));
    T->bytes = 2 * n * sizeof(uint32_t) + (size_t)n_yaw * (sizeof(p2p::YawDesc) + sizeof(double));
    std::vector<double> yr(n_yaw);
    for (int i = 0; i < n_yaw; ++i)
        yr[i] = deg2rad(yaw_deg[i]);  // P:85
    hipStream_t st = ctx->stream;
    StreamSyncGuard sync_on_exit(st);  // yr, the caller's rows and T's blocks outlive whatever an error return leaves queued
    HIP_TRY(hipMemcpyAsync(T->d_yaw_rad, yr.data(), yr.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(ctx->ev_t0, st));
    if (rows) {
        HIP_TRY(hipMemcpyAsync(d_rows, rows, n * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(p2p::launch_yaw_pack(T->d_ytab, d_rows, n, st));
    }
    // (without caller rows the descriptor kernel makes the packed table as it goes: one launch)
    HIP_TRY(p2p::launch_yaw_desc(T->d_ydesc, T->d_f4tab, T->d_ytab, pw, n_yaw, rows ? nullptr : T->d_yaw_rad, st));
    HIP_TRY(hipEventRecord(ctx->ev_t1, st));
    T->desc.resize(n_yaw);
    HIP_TRY(hipMemcpyAsync(T->desc.data(), T->d_ydesc, T->desc.size() * sizeof(p2p::YawDesc), hipMemcpyDeviceToHost, st));
    if (while_device_works)
        while_device_works();
    HIP_TRY(hipStreamSynchronize(st));  // yr (and the caller's rows) are stack-lifetime host buffers
    sync_on_exit.armed = false;
    (void)hipEventElapsedTime(&T->tables_ms, ctx->ev_t0, ctx->ev_t1);
    if (cached) {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        T->stamp = ++ctx->cache_clock;
        ctx->yaw_tabs[key] = T;
        cache_trim(ctx);
    }
    *out = T;
    return P2P_OK;
}

}  // namespace p2p_host
