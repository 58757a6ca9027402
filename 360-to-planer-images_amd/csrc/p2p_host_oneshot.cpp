// p2p_host_oneshot.cpp -- the one-shot slot pool and the host-buffer entry points on it (what the Python drop-in functions bind).
// Part of the host side of libp2p_hip.so (see p2p_host.h for the units); C ABI: include/p2p_hip.h via p2p_abi.cpp.
#include "p2p_host.h"

namespace p2p_host {

// The one-shot entry points run on a small pool of contexts per device (P2P_ONESHOT_SLOTS, default 4), not on
// one per calling thread: the reference fans process_yaw_and_pitchs out to int(0.9 * cores) threads (P:252-265,
// P:304-306) -- 230 on a 256-core host -- and one stream plus one cached job (a device copy of the panorama, the
// views, the plan) per thread would be hundreds of streams and tens of GB.  A caller takes an idle slot
// (preferring one whose cached job has its geometry: the reference keeps its maps for the life of the process,
// P:17-18), waits if all are busy, and gives it back.  The pool is never torn down at exit: no HIP call runs
// after the runtime's own shutdown, the OS reclaims the memory; release_cache() frees it on request.
struct OneShotSlot {
    int device = 0;
    bool busy = false;
    p2p_ctx* ctx = nullptr;
    p2p_job* cached = nullptr;
};

struct OneShotPool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<OneShotSlot*> slots;
};

OneShotPool& pool()
{
    static OneShotPool* p = new OneShotPool();  // intentionally never destroyed (see above)
    return *p;
}

// `wants(job)` says whether a slot's cached job can be re-used as is
template <class F>
int slot_acquire(int device, F wants, OneShotSlot** out)
{
    *out = nullptr;
    int rc = use_device(device);
    if (rc != P2P_OK)
        return rc;
    const int max_slots = options().oneshot_slots;
    OneShotPool& P = pool();
    std::unique_lock<std::mutex> lk(P.mu);
    for (;;) {
        OneShotSlot* idle = nullptr;
        int n_dev = 0;
        for (OneShotSlot* s : P.slots) {
            if (s->device != device)
                continue;
            ++n_dev;
            if (s->busy)
                continue;
            if (s->cached && wants(s->cached)) {
                idle = s;
                break;
            }
            if (!idle || (idle->cached && !s->cached))
                idle = s;  // otherwise prefer a slot that holds nothing
        }
        if (!idle && n_dev < max_slots) {
            idle = new (std::nothrow) OneShotSlot();
            if (!idle)
                return fail(P2P_ERR_OOM, "host allocation failed");
            idle->device = device;
            P.slots.push_back(idle);
        }
        if (idle) {
            idle->busy = true;
            lk.unlock();
            if (!idle->ctx) {
                rc = ctx_create(device, &idle->ctx);
                if (rc != P2P_OK) {
                    lk.lock();
                    idle->busy = false;
                    P.cv.notify_all();
                    return rc;
                }
            } else {
                (void)hipSetDevice(device);
            }
            *out = idle;
            return P2P_OK;
        }
        P.cv.wait(lk);
    }
}

void slot_release(OneShotSlot* s)
{
    OneShotPool& P = pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        s->busy = false;
    }
    P.cv.notify_all();  // one condition variable, waiters for several devices: the right one must wake
}

struct SlotGuard {  // gives the slot back on every return path
    OneShotSlot* s = nullptr;
    ~SlotGuard() { if (s) slot_release(s); }
};


int host_alloc(size_t bytes, void** out)
{
    if (!out)
        return fail(P2P_ERR_INVALID, "p2p_host_alloc: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(P2P_ERR_NO_DEVICE, "no HIP device is available (hipGetDeviceCount found none)");
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(e == hipErrorOutOfMemory ? P2P_ERR_OOM : P2P_ERR_HIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    return P2P_OK;
}

int host_free(void* ptr)
{
    if (!ptr)
        return P2P_OK;
    HIP_TRY(hipHostFree(ptr));
    return P2P_OK;
}

int release_cache(void)
{
    DeviceRestore keep;  // the calling thread stays on its device
    // every idle slot's cached job (busy ones belong to calls in flight on other threads)
    OneShotPool& P = pool();
    std::vector<p2p_job*> victims;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (OneShotSlot* s : P.slots)
            if (!s->busy) {
                if (s->cached) {
                    victims.push_back(s->cached);
                    s->cached = nullptr;
                }
                if (s->ctx) {
                    (void)hipSetDevice(s->device);
                    (void)hipStreamSynchronize(s->ctx->stream);
                    for (int i = 0; i < 4; ++i) {
                        (void)dev_free(s->ctx->scratch[i]);
                        s->ctx->scratch[i] = nullptr;
                        s->ctx->scratch_bytes[i] = 0;
                    }
                }
            }
    }
    for (p2p_job* j : victims)
        job_destroy(j);
    // the tables and plans no job uses any more, of EVERY live context (the slots' and the caller's own), then the
    // pool's idle blocks back to the driver
    (void)caches_evict_all();
    dev_pool_trim();
    pin_pool_trim();
    return P2P_OK;
}

int device_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes)
{
    if (!free_bytes || !total_bytes)
        return fail(P2P_ERR_INVALID, "NULL argument");
    DeviceRestore keep;
    int rc = use_device(device);
    if (rc != P2P_OK)
        return rc;
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f;
    *total_bytes = (int64_t)t;
    return P2P_OK;
}

int reload_options(void)
{
    options_reload();
    return P2P_OK;
}


// ------------------------------------------------------------------------------------------
// one-shot entry points
// ------------------------------------------------------------------------------------------
int views_oneshot(const uint8_t* pano, int pw, int ph, int64_t row_stride,
                         const double* yaw_deg, int n_yaw, const double* pitch_deg, int n_pitch,
                         double fov_deg, int ow, int oh, uint8_t* out, int device, int flags,
                         const float* yaw_rows, const float* U, const float* V, int border = 0, unsigned long long maps_key = 0)
{
    if (!pano || !out)
        return fail(P2P_ERR_INVALID, "NULL image pointer");
    if (n_yaw == 0 || n_pitch == 0)
        return P2P_OK;
    std::vector<double> dummy_yaw, dummy_pitch;
    if (!yaw_deg) {  // caller-supplied rows: degrees are irrelevant
        dummy_yaw.assign(n_yaw, 0.0);
        yaw_deg = dummy_yaw.data();
    }
    if (!pitch_deg) {
        dummy_pitch.assign(n_pitch, 90.0);
        pitch_deg = dummy_pitch.data();
    }
    p2p_job_desc_f64 d{};
    d.pw = pw; d.ph = ph; d.n_panos = 1;
    d.n_yaw = n_yaw; d.yaw_deg = yaw_deg;
    d.n_pitch = n_pitch; d.pitch_deg = pitch_deg;
    d.fov_deg = fov_deg; d.ow = ow; d.oh = oh; d.flags = flags;

    // a slot of the one-shot pool, preferably one whose cached job has this call's geometry
    const Options now = options();
    auto same_options = [&](const Options& a) {
        return a.plan_cache == now.plan_cache && a.tile_shape == now.tile_shape && a.pairs_per_block == now.pairs_per_block &&
               a.max_pairs_per_block == now.max_pairs_per_block && a.chunk_outer == now.chunk_outer && a.main_order == now.main_order &&
               a.main_group == now.main_group && a.main_span == now.main_span && a.main_tail == now.main_tail && a.main_tail_parts == now.main_tail_parts && a.prefetch_lead == now.prefetch_lead && a.force_rest == now.force_rest &&
               a.gather_ppb == now.gather_ppb && a.gather_all == now.gather_all && a.gather_blocky_from == now.gather_blocky_from &&
               a.gather_order == now.gather_order && a.gather_group == now.gather_group && a.scramble_plan == now.scramble_plan &&
               a.coords_all == now.coords_all && a.merge_gather == now.merge_gather && a.pair_ctx_table == now.pair_ctx_table &&
               a.early_main == now.early_main && a.defer_lists == now.defer_lists && a.band == now.band && a.band_bh == now.band_bh &&
               a.band_cw == now.band_cw && a.band_maxw == now.band_maxw && a.band_maxh == now.band_maxh;
    };
    auto same_geometry = [&](const p2p_job* c) {
        const p2p_job_desc& k = c->d;
        return same_options(c->opt) && k.pw == pw && k.ph == ph && k.n_yaw == n_yaw && k.n_pitch == n_pitch && c->fov == fov_deg &&
               k.ow == ow && k.oh == oh && k.flags == flags && c->border == border && c->host_maps == (U != nullptr) &&
               std::equal(c->pitch.begin(), c->pitch.end(), pitch_deg);
    };
    // (caller maps that carry a key: a slot whose job already holds exactly those maps -- and the plan made from them -- first)
    auto holds_maps = [&](const p2p_job* c) { return same_geometry(c) && (!U || maps_key == 0 || c->maps_key == maps_key); };
    SlotGuard guard;
    int rc = slot_acquire(device, holds_maps, &guard.s);
    if (rc != P2P_OK)
        return rc;
    OneShotSlot* slot = guard.s;
    p2p_job* j = nullptr;
    if (slot->cached) {
        p2p_job* c = slot->cached;
        if (same_geometry(c)) {
            j = c;
            // caller rows replace the tables below; otherwise rebuild them only when the yaws changed
            if (!yaw_rows && (c->rows_from_host || !std::equal(c->yaw.begin(), c->yaw.end(), yaw_deg)))
                rc = job_set_yaws_f64(c, yaw_deg);
        } else {
            job_destroy(c);
            slot->cached = nullptr;
        }
    }
    const bool fresh = (j == nullptr);
    if (fresh) {
        rc = job_create_f64(slot->ctx, &d, &j);
        if (rc != P2P_OK)
            return rc;
        j->border = border;
    }
    // The job is this call's until it ends well: on an error return or an exception (a host allocation failure anywhere
    // below) it is destroyed, never left behind in the slot half-updated.
    slot->cached = nullptr;
    struct Owner {
        p2p_job* j;
        ~Owner() { if (j) job_destroy(j); }
    } owner{j};
    if (rc == P2P_OK)
        rc = job_set_pano(j, 0, pano, row_stride);
    if (rc == P2P_OK && U && !(maps_key != 0 && j->host_maps && j->maps_key == maps_key && !yaw_rows)) {
        j->maps_key = 0;
        rc = job_set_maps(j, yaw_rows, U, V);
        if (rc == P2P_OK)
            j->maps_key = maps_key;
    }
    if (rc == P2P_OK)
        rc = job_run(j);
    if (rc == P2P_OK)
        rc = job_get_views(j, 0, out);

    const size_t held = j->pano_stride + j->out_bytes + (size_t)n_yaw * pw * 8 +
                        (size_t)n_pitch * ow * oh * (U ? 28 : 20);
    const bool keep = rc == P2P_OK && j->opt.oneshot_cache != 0 && held <= (size_t)j->opt.oneshot_cache_max_mb * 1048576ull;
    if (keep) {
        slot->cached = j;
        owner.j = nullptr;
    }
    return rc;
}

int remap_views_f64(const uint8_t* pano, int pw, int ph, int64_t row_stride,
                        const double* yaw_deg, int n_yaw, const double* pitch_deg, int n_pitch,
                        double fov_deg, int ow, int oh, uint8_t* out, int device, int flags)
{
    if (n_yaw < 0 || n_pitch < 0 || (n_yaw > 0 && !yaw_deg) || (n_pitch > 0 && !pitch_deg))
        return fail(P2P_ERR_INVALID, "bad yaw/pitch list");
    return views_oneshot(pano, pw, ph, row_stride, yaw_deg, n_yaw, pitch_deg, n_pitch, fov_deg, ow, oh,
                         out, device, flags, nullptr, nullptr, nullptr);
}

int remap_views_u8(const uint8_t* pano, int pw, int ph, int64_t row_stride,
                       const int32_t* yaw_deg, int n_yaw, const int32_t* pitch_deg, int n_pitch,
                       int fov_deg, int ow, int oh, uint8_t* out, int device, int flags)
{
    if (n_yaw < 0 || n_pitch < 0 || (n_yaw > 0 && !yaw_deg) || (n_pitch > 0 && !pitch_deg))
        return fail(P2P_ERR_INVALID, "bad yaw/pitch list");
    // the integer entry point keeps the CLI's validation (check_pitch, P:362-376)
    for (int i = 0; i < n_pitch; ++i)
        if (pitch_deg[i] < 1 || pitch_deg[i] > 179)
            return fail(P2P_ERR_INVALID, "Pitch angle must be between 1 and 179 degrees, got %d.", pitch_deg[i]);
    std::vector<double> yaw(yaw_deg, yaw_deg + n_yaw), pitch(pitch_deg, pitch_deg + n_pitch);
    return remap_views_f64(pano, pw, ph, row_stride, yaw.data(), n_yaw, pitch.data(), n_pitch, (double)fov_deg,
                               ow, oh, out, device, flags);
}

int remap_views_maps_u8(const uint8_t* pano, int pw, int ph, int64_t row_stride,
                            const float* yaw_rows, int n_yaw, const float* U, const float* V,
                            int n_pitch, int ow, int oh, uint8_t* out, int device)
{
    if (n_yaw < 0 || n_pitch < 0 || (n_yaw > 0 && !yaw_rows) || (n_pitch > 0 && (!U || !V)))
        return fail(P2P_ERR_INVALID, "bad map arguments");
    return views_oneshot(pano, pw, ph, row_stride, nullptr, n_yaw, nullptr, n_pitch, 90.0, ow, oh, out,
                         device, 0, yaw_rows, U, V);
}

int remap_views_pitch_maps_f64(const uint8_t* pano, int pw, int ph, int64_t row_stride,
                                   const double* yaw_deg, int n_yaw, const float* U, const float* V, int n_pitch,
                                   uint64_t maps_key, int ow, int oh, uint8_t* out, int device)
{
    if (n_yaw < 0 || n_pitch < 0 || (n_yaw > 0 && !yaw_deg) || (n_pitch > 0 && (!U || !V)))
        return fail(P2P_ERR_INVALID, "bad yaw list / map arguments");
    return views_oneshot(pano, pw, ph, row_stride, yaw_deg, n_yaw, nullptr, n_pitch, 90.0, ow, oh, out,
                         device, 0, nullptr, U, V, 0, (unsigned long long)maps_key);
}

int remap_maps_u8(const uint8_t* src, int sw, int sh, int64_t row_stride, int cn,
                      const float* U, const float* V, int ow, int oh, uint8_t* out,
                      int border_mode, const uint8_t* border_value, int device)
{
    return remap_maps_interp_u8(src, sw, sh, row_stride, cn, U, V, ow, oh, out, P2P_INTER_LINEAR,
                                    border_mode, border_value, device);
}

int remap_maps_batch_u8(const uint8_t* src, int sw, int sh, int64_t row_stride,
                            const float* U, const float* V, int n_maps, int ow, int oh, uint8_t* out,
                            int border_mode, int device)
{
    if (!src || !U || !V || !out)
        return fail(P2P_ERR_INVALID, "NULL pointer");
    if (n_maps < 0)
        return fail(P2P_ERR_INVALID, "bad map count");
    if (border_mode < P2P_BORDER_CONSTANT || border_mode > P2P_BORDER_REFLECT_101)
        return fail(P2P_ERR_INVALID, "unsupported border mode %d", border_mode);
    // the view kernel with an identity yaw stage and the n_maps caller maps as its "pitch" views: one upload of the
    // image, one plan pass, one launch for all of them
    const double yaw0 = 0.0;
    return views_oneshot(src, sw, sh, row_stride, &yaw0, 1, nullptr, n_maps, 90.0, ow, oh, out, device, 0,
                         nullptr, U, V, border_mode);
}

int remap_maps_interp_u8(const uint8_t* src, int sw, int sh, int64_t row_stride, int cn,
                             const float* U, const float* V, int ow, int oh, uint8_t* out,
                             int interpolation, int border_mode, const uint8_t* border_value, int device)
{
    if (interpolation != P2P_INTER_NEAREST && interpolation != P2P_INTER_LINEAR && interpolation != P2P_INTER_CUBIC)
        return fail(P2P_ERR_INVALID, "unsupported interpolation %d", interpolation);
    if (!src || !U || !V || !out)
        return fail(P2P_ERR_INVALID, "NULL pointer");
    if (cn != 1 && cn != 3 && cn != 4)
        return fail(P2P_ERR_INVALID, "cn must be 1, 3 or 4 (got %d)", cn);
    if (!dims_ok(sw, sh) || !dims_ok(ow, oh))
        return fail(P2P_ERR_INVALID, "image sides must be in 1..32766 (cv::remap asserts < SHRT_MAX)");
    if (border_mode < P2P_BORDER_CONSTANT || border_mode > P2P_BORDER_REFLECT_101)
        return fail(P2P_ERR_INVALID, "unsupported border mode %d", border_mode);
    if (row_stride < (int64_t)sw * cn)
        return fail(P2P_ERR_INVALID, "row_stride too small");
    bool zero_border = true;
    for (int k = 0; k < cn; ++k)
        zero_border = zero_border && (!border_value || border_value[k] == 0);
    if (interpolation == P2P_INTER_LINEAR && cn == 3 && (border_mode != P2P_BORDER_CONSTANT || zero_border)) {
        // three interleaved channels: the view kernel with an identity yaw stage (yaw 0 quantises to
        // "column x, fraction 0", so stage 1 is a copy) and the caller's maps as its pitch stage --
        // LDS-staged taps instead of per-pixel byte gathers
        const double yaw0 = 0.0;
        return views_oneshot(src, sw, sh, row_stride, &yaw0, 1, nullptr, 1, 90.0, ow, oh, out, device, 0,
                             nullptr, U, V, border_mode);
    }
    SlotGuard guard;
    int rc = slot_acquire(device, [](const p2p_job*) { return false; }, &guard.s);
    if (rc != P2P_OK)
        return rc;
    p2p_ctx* ctx = guard.s->ctx;
    const int pitch = (sw * cn + 15) & ~15;
    const size_t n_map = (size_t)ow * oh;
    hipError_t e = hipSuccess;
    auto scratch = [&](int i, size_t bytes) -> void* {
        if (e == hipSuccess && ctx->scratch_bytes[i] < bytes) {
            (void)hipStreamSynchronize(ctx->stream);  // an earlier call that failed half-way may have left work in flight
            (void)dev_free(ctx->scratch[i]);
            ctx->scratch[i] = nullptr;
            ctx->scratch_bytes[i] = 0;
            e = dev_alloc(&ctx->scratch[i], bytes);
            if (e == hipSuccess)
                ctx->scratch_bytes[i] = bytes;
        }
        return ctx->scratch[i];
    };
    if (interpolation == P2P_INTER_CUBIC && !ctx->d_ctab) {
        e = dev_alloc((void**)&ctx->d_ctab, 1024 * 16 * sizeof(short));
        if (e == hipSuccess) e = p2p::launch_cubic_tab(ctx->d_ctab, ctx->stream);
    }
    uint8_t* d_src = (uint8_t*)scratch(0, (size_t)pitch * sh + kSlack);
    uint8_t* d_dst = (uint8_t*)scratch(1, n_map * cn);
    float* d_U = (float*)scratch(2, n_map * sizeof(float));
    float* d_V = (float*)scratch(3, n_map * sizeof(float));
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(d_src, pitch, src, (size_t)row_stride, (size_t)sw * cn, sh, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_U, U, n_map * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_V, V, n_map * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        p2p::RemapParams P{};
        P.src = d_src; P.sw = sw; P.sh = sh; P.src_pitch = pitch;
        P.U = d_U; P.V = d_V; P.dst = d_dst; P.ow = ow; P.oh = oh; P.border = border_mode;
        for (int k = 0; k < 4; ++k)
            P.cval[k] = (border_value && k < cn) ? border_value[k] : 0;
        P.ctab = ctx->d_ctab;
        e = p2p::launch_remap_maps(P, cn, interpolation, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_dst, n_map * cn, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);  // the caller's buffers are its own again when this returns
        return fail(e == hipErrorOutOfMemory ? P2P_ERR_OOM : P2P_ERR_HIP, "p2p_remap_maps_u8: %s", hipGetErrorString(e));
    }
    return P2P_OK;
}

int build_pitch_map(int ow, int oh, double fov_rad, double pitch_rad, int pw, int ph,
                        float* U, float* V, int device)
{
    if (!U || !V)
        return fail(P2P_ERR_INVALID, "NULL pointer");
    if (!dims_ok(ow, oh) || pw < 1 || ph < 1)
        return fail(P2P_ERR_INVALID, "bad sizes");
    SlotGuard guard;
    int rc = slot_acquire(device, [](const p2p_job*) { return false; }, &guard.s);
    if (rc != P2P_OK)
        return rc;
    p2p_ctx* ctx = guard.s->ctx;
    p2p::MapGeom g{};
    g.half_w = (float)(ow / 2.0);
    g.half_h = (float)(oh / 2.0);
    g.focal = (float)((0.5 * ow) / std::tan(fov_rad / 2));  // P:119, cast to float32 at P:131
    g.pw_f = (float)pw;
    g.ph_f = (float)ph;
    const double pr = pitch_rad;
    const size_t n = (size_t)ow * oh;
    float *dU = nullptr, *dV = nullptr;
    hipError_t e = dev_alloc((void**)&dU, n * sizeof(float));
    if (e == hipSuccess) e = dev_alloc((void**)&dV, n * sizeof(float));
    if (e == hipSuccess) e = p2p::launch_pitch_map(dU, dV, ow, oh, g, (float)std::cos(pr), (float)std::sin(pr), ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(U, dU, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(V, dV, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) (void)hipStreamSynchronize(ctx->stream);  // nothing queued may still touch the blocks freed below
    (void)dev_free(dU);
    (void)dev_free(dV);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? P2P_ERR_OOM : P2P_ERR_HIP, "p2p_build_pitch_map: %s", hipGetErrorString(e));
    return P2P_OK;
}

int build_rot_map(int ow, int oh, double fov_rad, const float* R9, int pw, int ph,
                      float* U, float* V, int device)
{
    if (!U || !V || !R9)
        return fail(P2P_ERR_INVALID, "NULL pointer");
    if (!dims_ok(ow, oh) || pw < 1 || ph < 1)
        return fail(P2P_ERR_INVALID, "bad sizes");
    SlotGuard guard;
    int rc = slot_acquire(device, [](const p2p_job*) { return false; }, &guard.s);
    if (rc != P2P_OK)
        return rc;
    p2p_ctx* ctx = guard.s->ctx;
    p2p::MapGeom g{};
    g.half_w = (float)(ow / 2.0);
    g.half_h = (float)(oh / 2.0);
    g.focal = (float)((0.5 * ow) / std::tan(fov_rad / 2));  // L:95, cast to float32 at L:109
    g.pw_f = (float)pw;
    g.ph_f = (float)ph;
    const size_t n = (size_t)ow * oh;
    float *dU = nullptr, *dV = nullptr;
    hipError_t e = dev_alloc((void**)&dU, n * sizeof(float));
    if (e == hipSuccess) e = dev_alloc((void**)&dV, n * sizeof(float));
    if (e == hipSuccess) e = p2p::launch_rot_map(dU, dV, ow, oh, g, R9, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(U, dU, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(V, dV, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) (void)hipStreamSynchronize(ctx->stream);  // nothing queued may still touch the blocks freed below
    (void)dev_free(dU);
    (void)dev_free(dV);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? P2P_ERR_OOM : P2P_ERR_HIP, "p2p_build_rot_map: %s", hipGetErrorString(e));
    return P2P_OK;
}

int build_yaw_row(int pw, double yaw_rad, float* U_row, int device)
{
    if (!U_row)
        return fail(P2P_ERR_INVALID, "NULL pointer");
    if (pw < 1 || pw >= 32767)
        return fail(P2P_ERR_INVALID, "bad panorama width %d", pw);
    SlotGuard guard;
    int rc = slot_acquire(device, [](const p2p_job*) { return false; }, &guard.s);
    if (rc != P2P_OK)
        return rc;
    p2p_ctx* ctx = guard.s->ctx;
    const double yr = yaw_rad;  // np.radians(yaw_angle), P:85
    double* d_yr = nullptr;
    float* d_row = nullptr;
    hipError_t e = dev_alloc((void**)&d_yr, sizeof(double));
    if (e == hipSuccess) e = dev_alloc((void**)&d_row, (size_t)pw * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_yr, &yr, sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = p2p::launch_yaw_tables(nullptr, d_row, pw, 1, d_yr, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(U_row, d_row, (size_t)pw * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) (void)hipStreamSynchronize(ctx->stream);  // nothing queued may still touch the blocks freed below
    (void)dev_free(d_yr);
    (void)dev_free(d_row);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? P2P_ERR_OOM : P2P_ERR_HIP, "p2p_build_yaw_row: %s", hipGetErrorString(e));
    return P2P_OK;
}

}  // namespace p2p_host
