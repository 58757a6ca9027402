// p2p_host_job.cpp -- jobs: resident panoramas and views, uploads, p2p_job_run's launch sequence, downloads, launch timing.
// Part of the host side of libp2p_hip.so (see p2p_host.h for the units); C ABI: include/p2p_hip.h via p2p_abi.cpp.
#include "p2p_host.h"

namespace p2p_host {

void job_destroy(p2p_job* j)
{
    if (!j)
        return;
    if (j->ctx) {
        (void)hipSetDevice(j->ctx->device);
        for (hipStream_t st : {j->ctx->stream_up, j->ctx->stream, j->ctx->stream_down})
            if (st) (void)hipStreamSynchronize(st);  // nothing queued touches the blocks that go back to the pool below
    }
    for (hipEvent_t e : {j->ev_up, j->ev_run, j->ev_down})
        if (e) (void)hipEventDestroy(e);
    if (j->owns_src)
        (void)dev_free(j->d_src);
    (void)dev_free(j->d_out);
    (void)dev_free(j->d_pack);
    j->plan_ref.reset();  // the tables themselves go when the last job and the context's cache let go of them
    j->yaw_ref.reset();
    (void)dev_free(j->d_pitch);
    (void)dev_free(j->d_mapU);
    (void)dev_free(j->d_mapV);
    (void)dev_free(j->d_rows);
    (void)dev_free(j->d_odd_pairs);
    (void)dev_free(j->d_pitch_order);
    (void)dev_free(j->d_view_mask);
    (void)dev_free(j->d_pair_ctx);
    for (hipEvent_t e : j->ev_ring)
        (void)hipEventDestroy(e);
    delete j;
}


int job_create_core(p2p_ctx* ctx, const p2p_job_desc& d, const double* yaw_deg, const double* pitch_deg,
                           double fov_deg, p2p_job** out)
{
    *out = nullptr;
    if (!dims_ok(d.pw, d.ph))
        return fail(P2P_ERR_INVALID, "panorama %dx%d: both sides must be in 1..32766 (cv::remap asserts < SHRT_MAX)", d.pw, d.ph);
    if (!dims_ok(d.ow, d.oh))
        return fail(P2P_ERR_INVALID, "output %dx%d: both sides must be in 1..32766", d.ow, d.oh);
    if (d.n_panos < 1 || d.n_yaw < 1 || d.n_pitch < 1 || !yaw_deg || !pitch_deg)
        return fail(P2P_ERR_INVALID, "need at least one panorama, yaw and pitch");
    if (d.n_pitch > 65535 || d.n_yaw > 65535)
        return fail(P2P_ERR_INVALID, "at most 65535 pitch angles and 65535 yaw angles per job (got %d, %d)", d.n_pitch, d.n_yaw);
    if ((d.flags & P2P_FLAG_PIXEL_CENTRES) && !(d.flags & (P2P_FLAG_PIXELS_F32 | P2P_FLAG_PIXELS_F16)))
        return fail(P2P_ERR_INVALID, "P2P_FLAG_PIXEL_CENTRES needs one of the float pixel paths (the uint8 path is the reference's arithmetic)");
    if (d.n_panos >= (1 << 26))
        return fail(P2P_ERR_INVALID, "at most 2^26 - 1 panoramas per job");
    if ((unsigned long long)d.n_panos * d.n_yaw * d.n_yaw >= (1ull << 32))
        return fail(P2P_ERR_INVALID, "n_panos * n_yaw^2 must stay below 2^32 (got %d panoramas, %d yaws)", d.n_panos, d.n_yaw);
    if (!std::isfinite(fov_deg))
        return fail(P2P_ERR_INVALID, "FOV must be a finite number of degrees");
    for (int i = 0; i < d.n_yaw; ++i)
        if (!std::isfinite(yaw_deg[i]))
            return fail(P2P_ERR_INVALID, "yaw angle %d is not finite", i);
    for (int i = 0; i < d.n_pitch; ++i)
        if (!std::isfinite(pitch_deg[i]))
            return fail(P2P_ERR_INVALID, "pitch angle %d is not finite", i);
    HIP_TRY(hipSetDevice(ctx->device));

    p2p_job* j = new (std::nothrow) p2p_job();
    if (!j)
        return fail(P2P_ERR_OOM, "host allocation failed");
    // (a half-made job goes on every path that does not hand it over: error returns, and exceptions -- the vectors below)
    struct Owner {
        p2p_job* j;
        ~Owner() { if (j) job_destroy(j); }
    } owner{j};
    j->ctx = ctx;
    j->opt = options();
    j->d = d;
    j->d.yaw_deg = nullptr;
    j->d.pitch_deg = nullptr;
    j->yaw.assign(yaw_deg, yaw_deg + d.n_yaw);
    j->pitch.assign(pitch_deg, pitch_deg + d.n_pitch);
    j->fov = fov_deg;
    j->pano_set.assign(d.n_panos, 0);
    j->n_views_wanted = d.n_yaw * d.n_pitch;

    j->src_pitch = (3 * (d.pw + p2p::PANO_PAD) + 15) & ~15;  // every row is followed by a copy of its first pixels
    j->pano_stride = (((size_t)j->src_pitch * d.ph + kSlack) + 255) & ~(size_t)255;
    j->out_row = 12 * ((d.ow + 3) / 4);  // whole 4-pixel groups: every row starts dword-aligned, whatever the width
    j->out_bytes = (size_t)d.n_panos * d.n_yaw * d.n_pitch * d.oh * j->out_row;

    // scalars NumPy evaluates in float64 once per map (P:64-68, P:119, P:129-131, P:142-149)
    const double fov_rad = deg2rad(fov_deg);
    j->geom.half_w = (float)(d.ow / 2.0);
    j->geom.half_h = (float)(d.oh / 2.0);
    j->geom.focal = (float)((0.5 * d.ow) / std::tan(fov_rad / 2));
    j->geom.pw_f = (float)d.pw;
    j->geom.ph_f = (float)d.ph;
    std::vector<p2p::PitchConst> pc(d.n_pitch);
    for (int i = 0; i < d.n_pitch; ++i) {
        double pr = deg2rad(j->pitch[i]);
        pc[i].c = (float)std::cos(pr);
        pc[i].s = (float)std::sin(pr);
    }

    hipError_t e = dev_alloc((void**)&j->d_src, j->pano_stride * d.n_panos);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&j->ev_up, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&j->ev_run, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&j->ev_down, hipEventDisableTiming);
    if (e == hipSuccess) e = dev_alloc((void**)&j->d_out, j->out_bytes + 16);
    if (e == hipSuccess) e = dev_alloc((void**)&j->d_pitch, (size_t)d.n_pitch * sizeof(p2p::PitchConst));
    j->row0 = 0;
    j->row1 = d.oh;
    j->shape = choose_shape(d, j->opt);
    {
        const p2p::TileShape& S = shape_ops(j->shape).shape;
        j->n_tiles = (size_t)((d.ow + S.tile_w - 1) / S.tile_w) * ((d.oh + S.tile_h - 1) / S.tile_h);
    }
    {
        const size_t slots = j->n_tiles * d.n_pitch;
        if (slots >= 0x7FFFFFFFull) {
            return fail(P2P_ERR_INVALID, "too many tiles (%zu): fewer pitch angles or smaller views per job", slots);
        }
        if (e == hipSuccess) e = dev_alloc((void**)&j->d_pitch_order, (size_t)d.n_pitch * sizeof(uint16_t));
        // views looking further from the horizon have larger source footprints: launch them first
        std::vector<uint16_t> ord(d.n_pitch);
        for (int i = 0; i < d.n_pitch; ++i)
            ord[i] = (uint16_t)i;
        std::stable_sort(ord.begin(), ord.end(), [&](uint16_t a, uint16_t b) {
            return std::fabs(j->pitch[a] - 90.0) > std::fabs(j->pitch[b] - 90.0);
        });
        if (e == hipSuccess)
            e = hipMemcpy(j->d_pitch_order, ord.data(), ord.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(j->d_pitch, pc.data(), pc.size() * sizeof(p2p::PitchConst), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);  // pc is a stack-lifetime host buffer
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? P2P_ERR_OOM : P2P_ERR_HIP, "p2p_job_create: %s", hipGetErrorString(e));
    // the yaw tables: the context's, if it has built them for these angles before (P:42-52)
    std::shared_ptr<YawTabs> yt;
    // (while the device makes them, the host fetches the block the job's first run will want for its plan)
    int rc = yaw_tabs_get(ctx, d.pw, j->yaw, nullptr, nullptr, j->opt.plan_cache != 0, &yt, [j]() { plan_block_prefetch(j); });
    if (rc == P2P_OK)
        rc = job_adopt_yaw_tabs(j, yt);
    if (rc != P2P_OK)
        return rc;
    owner.j = nullptr;
    *out = j;
    return P2P_OK;
}

int job_create(p2p_ctx* ctx, const p2p_job_desc* desc, p2p_job** out)
{
    if (!ctx || !desc || !out)
        return fail(P2P_ERR_INVALID, "p2p_job_create: NULL argument");
    *out = nullptr;
    const p2p_job_desc& d = *desc;
    if (d.n_yaw < 1 || d.n_pitch < 1 || !d.yaw_deg || !d.pitch_deg)
        return fail(P2P_ERR_INVALID, "need at least one panorama, yaw and pitch");
    // the integer entry point keeps the CLI's validation (check_pitch, P:362-376)
    for (int i = 0; i < d.n_pitch; ++i)
        if (d.pitch_deg[i] < 1 || d.pitch_deg[i] > 179)
            return fail(P2P_ERR_INVALID, "Pitch angle must be between 1 and 179 degrees, got %d.", d.pitch_deg[i]);
    std::vector<double> yaw(d.yaw_deg, d.yaw_deg + d.n_yaw), pitch(d.pitch_deg, d.pitch_deg + d.n_pitch);
    return job_create_core(ctx, d, yaw.data(), pitch.data(), (double)d.fov_deg, out);
}

int job_create_f64(p2p_ctx* ctx, const p2p_job_desc_f64* desc, p2p_job** out)
{
    if (!ctx || !desc || !out)
        return fail(P2P_ERR_INVALID, "p2p_job_create_f64: NULL argument");
    p2p_job_desc d{};
    d.pw = desc->pw; d.ph = desc->ph; d.n_panos = desc->n_panos;
    d.n_yaw = desc->n_yaw; d.n_pitch = desc->n_pitch;
    d.fov_deg = (int32_t)std::lround(std::isfinite(desc->fov_deg) ? desc->fov_deg : 0.0);
    d.ow = desc->ow; d.oh = desc->oh; d.flags = desc->flags;
    return job_create_core(ctx, d, desc->yaw_deg, desc->pitch_deg, desc->fov_deg, out);
}

// ev_run := "everything enqueued on the kernel stream so far", which covers the job's last run; recorded lazily,
// when a copy needs the ordering, so that back-to-back launches pay nothing for it
int mark_run(p2p_job* j)
{
    if (j->run_unmarked) {
        HIP_TRY(hipEventRecord(j->ev_run, j->ctx->stream));
        j->ev_run_recorded = true;
        j->run_unmarked = false;
    }
    return P2P_OK;
}

// the two copies of one panorama upload: the rows, and behind every row a copy of its first pixels (PANO_PAD of them,
// or the whole row if it is shorter) -- the gather kernel reads the pixels under a yaw shift that runs across the
// row's end as one contiguous run
int enqueue_pano_copy(p2p_job* j, int index, const uint8_t* pano, int64_t row_stride, hipStream_t st)
{
    uint8_t* dst = j->d_src + (size_t)index * j->pano_stride;
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)j->src_pitch, pano, (size_t)row_stride, (size_t)3 * j->d.pw, (size_t)j->d.ph,
                             hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(dst + (size_t)3 * j->d.pw, (size_t)j->src_pitch, pano, (size_t)row_stride,
                             (size_t)3 * std::min(j->d.pw, p2p::PANO_PAD), (size_t)j->d.ph, hipMemcpyHostToDevice, st));
    return P2P_OK;
}

int set_pano_check(p2p_job* j, int index, const uint8_t* pano, int64_t row_stride)
{
    if (!j || !pano)
        return fail(P2P_ERR_INVALID, "p2p_job_set_pano: NULL argument");
    if (!j->owns_src)
        return fail(P2P_ERR_STATE, "this job borrows its panoramas (p2p_job_share_panos): set them on the owning job");
    if (index < 0 || index >= j->d.n_panos)
        return fail(P2P_ERR_INVALID, "panorama index %d out of range", index);
    if (row_stride < (int64_t)3 * j->d.pw)
        return fail(P2P_ERR_INVALID, "row_stride %lld < 3*pw", (long long)row_stride);
    return P2P_OK;
}

int job_set_pano_async(p2p_job* j, int index, const uint8_t* pano, int64_t row_stride)
{
    if (int rc = set_pano_check(j, index, pano, row_stride))
        return rc;
    HIP_TRY(hipSetDevice(j->ctx->device));
    hipStream_t up = nullptr;
    HIP_TRY(ctx_copy_stream(j->ctx, true, &up));
    // on the upload stream, behind the last kernel that reads this job's panoramas: the copy overlaps whatever
    // other jobs of the context are running (the driver keeps two jobs per device and alternates)
    if (int rc = mark_run(j))
        return rc;
    if (j->ev_run_recorded)
        HIP_TRY(hipStreamWaitEvent(up, j->ev_run, 0));
    if (int rc = enqueue_pano_copy(j, index, pano, row_stride, up))
        return rc;
    HIP_TRY(hipEventRecord(j->ev_up, up));
    j->up_pending = true;
    j->pano_set[index] = 1;
    return P2P_OK;
}


int job_set_pano(p2p_job* j, int index, const uint8_t* pano, int64_t row_stride)
{
    if (int rc = set_pano_check(j, index, pano, row_stride))
        return rc;
    HIP_TRY(hipSetDevice(j->ctx->device));
    // in order on the kernel stream (behind every launch that reads the panoramas, ahead of the next one): no second
    // hardware queue for callers that never overlap copies with kernels
    StreamSyncGuard sync_on_exit(j->ctx->stream);  // the caller may release `pano` when we return, also on an error
    // (behind an asynchronous upload of the same job that is still in flight on the upload stream: two writers of
    // one panorama in unknown order otherwise)
    if (j->up_pending) {
        if (hipEventQuery(j->ev_up) != hipSuccess)
            HIP_TRY(hipStreamWaitEvent(j->ctx->stream, j->ev_up, 0));
        j->up_pending = false;
    }
    if (int rc = enqueue_pano_copy(j, index, pano, row_stride, j->ctx->stream))
        return rc;
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));
    sync_on_exit.armed = false;
    j->pano_set[index] = 1;
    return P2P_OK;
}

int job_share_panos(p2p_job* j, p2p_job* owner)
{
    if (!j || !owner || j == owner)
        return fail(P2P_ERR_INVALID, "p2p_job_share_panos: bad argument");
    if (!owner->owns_src)
        owner = owner->src_owner;
    if (j->ctx != owner->ctx || j->d.pw != owner->d.pw || j->d.ph != owner->d.ph || j->d.n_panos != owner->d.n_panos)
        return fail(P2P_ERR_INVALID, "jobs that share panoramas need one context, one panorama size and one panorama count");
    HIP_TRY(hipSetDevice(j->ctx->device));
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));
    if (j->ctx->stream_up)
        HIP_TRY(hipStreamSynchronize(j->ctx->stream_up));  // an asynchronous upload into the block that goes back to the pool
    j->up_pending = false;
    if (j->owns_src)
        (void)dev_free(j->d_src);
    j->d_src = owner->d_src;
    j->owns_src = false;
    j->src_owner = owner;
    return P2P_OK;
}

int job_set_yaws_f64(p2p_job* j, const double* yaw_deg)
{
    if (!j || !yaw_deg)
        return fail(P2P_ERR_INVALID, "p2p_job_set_yaws: NULL argument");
    const p2p_job_desc& d = j->d;
    for (int i = 0; i < d.n_yaw; ++i)
        if (!std::isfinite(yaw_deg[i]))
            return fail(P2P_ERR_INVALID, "yaw angle %d is not finite", i);
    HIP_TRY(hipSetDevice(j->ctx->device));
    j->yaw.assign(yaw_deg, yaw_deg + d.n_yaw);
    std::shared_ptr<YawTabs> yt;
    if (int rc = yaw_tabs_get(j->ctx, d.pw, j->yaw, nullptr, nullptr, j->opt.plan_cache != 0, &yt))
        return rc;
    j->rows_from_host = false;
    return job_adopt_yaw_tabs(j, yt);
}

int job_set_yaws(p2p_job* j, const int32_t* yaw_deg)
{
    if (!j || !yaw_deg)
        return fail(P2P_ERR_INVALID, "p2p_job_set_yaws: NULL argument");
    std::vector<double> y(yaw_deg, yaw_deg + j->d.n_yaw);
    return job_set_yaws_f64(j, y.data());
}

int job_set_maps(p2p_job* j, const float* yaw_rows, const float* U, const float* V)
{
    if (!j || !U || !V)
        return fail(P2P_ERR_INVALID, "p2p_job_set_maps: NULL argument");
    const p2p_job_desc& d = j->d;
    HIP_TRY(hipSetDevice(j->ctx->device));
    const size_t n_map = (size_t)d.n_pitch * d.oh * d.ow;
    if (yaw_rows) {
        // the yaw stage's taps must stay inside the row, as P:105's clip guarantees (checked before anything is
        // enqueued: an error return leaves no copy from the caller's buffers in flight and the job as it was)
        const size_t n = (size_t)d.n_yaw * d.pw;
        for (size_t k = 0; k < n; ++k)
            if (!(yaw_rows[k] >= 0.0f && yaw_rows[k] <= (float)(d.pw - 1)))
                return fail(P2P_ERR_INVALID, "yaw_rows[%zu] = %g outside [0, pw-1] (P:105 clips it)", k, (double)yaw_rows[k]);
    }
    if (!j->d_mapU) HIP_TRY(dev_alloc((void**)&j->d_mapU, n_map * sizeof(float)));
    if (!j->d_mapV) HIP_TRY(dev_alloc((void**)&j->d_mapV, n_map * sizeof(float)));
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));  // no launch in flight reads the plan that is about to go
    j->plan_ref.reset();  // the plan follows the maps (also when a later step of this call fails): a private one is built
    j->maps_key = 0;      // (whatever name the old maps had)
    {
        StreamSyncGuard sync_on_exit(j->ctx->stream);  // U and V are the caller's: nothing may still read them after a return
        HIP_TRY(hipMemcpyAsync(j->d_mapU, U, n_map * sizeof(float), hipMemcpyHostToDevice, j->ctx->stream));
        HIP_TRY(hipMemcpyAsync(j->d_mapV, V, n_map * sizeof(float), hipMemcpyHostToDevice, j->ctx->stream));
        HIP_TRY(hipStreamSynchronize(j->ctx->stream));
        sync_on_exit.armed = false;
    }
    j->host_maps = true;
    if (yaw_rows) {
        const size_t n = (size_t)d.n_yaw * d.pw;
        if (!j->d_rows) HIP_TRY(dev_alloc((void**)&j->d_rows, n * sizeof(float)));
        std::shared_ptr<YawTabs> yt;  // private tables: caller rows have no key
        if (int rc = yaw_tabs_get(j->ctx, d.pw, j->yaw, yaw_rows, j->d_rows, false, &yt))
            return rc;
        j->rows_from_host = true;
        return job_adopt_yaw_tabs(j, yt);
    }
    return P2P_OK;
}


// One image's ROWS shared out to several GPUs (every rank draws all views, a band of rows of each: a tile's set-up is
// then spread over all the pairs again, and the number of views no longer caps the speed-up).  Whole tile rows; the plan
// is made for the range (tiles outside it: mode 0, no kernel's), so the next run builds or fetches another plan.
int job_set_rows(p2p_job* j, int row0, int row1)
{
    if (!j)
        return fail(P2P_ERR_INVALID, "job is NULL");
    const int th = shape_ops(j->shape).shape.tile_h;
    if (row0 < 0 || row1 <= row0 || row1 > j->d.oh || row0 % th != 0 || (row1 % th != 0 && row1 != j->d.oh))
        return fail(P2P_ERR_INVALID, "rows [%d, %d) of %d: whole tile rows of %d (the last one may be short)", row0, row1, j->d.oh, th);
    if (row0 == j->row0 && row1 == j->row1)
        return P2P_OK;
    HIP_TRY(hipSetDevice(j->ctx->device));
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));  // no launch in flight reads the plan that is about to go
    j->row0 = row0;
    j->row1 = row1;
    j->plan_ref.reset();
    j->pc_plan = nullptr;  // (the pair-context table follows the plan's headers: rebuilt, whatever address the next plan gets)
    return P2P_OK;
}

// The border mode of the job's pitch stage (default BORDER_CONSTANT 0, the current tool's, P:212-218): the legacy tool's
// cv2.remap(..., borderMode=BORDER_REFLECT) (L:179) as a RESIDENT job -- its maps set once (p2p_job_set_maps), one upload
// and one launch per image.  The plan follows the mode (another key: the next run builds or fetches it).
int job_set_border(p2p_job* j, int border_mode)
{
    if (!j)
        return fail(P2P_ERR_INVALID, "job is NULL");
    if (border_mode < P2P_BORDER_CONSTANT || border_mode > P2P_BORDER_REFLECT_101)
        return fail(P2P_ERR_INVALID, "unsupported border mode %d", border_mode);
    if (j->d.flags & (P2P_FLAG_PIXELS_F32 | P2P_FLAG_PIXELS_F16))
        return fail(P2P_ERR_STATE, "the float pixel path wraps around the seam by itself: it has no border mode");
    if (border_mode == j->border)
        return P2P_OK;
    HIP_TRY(hipSetDevice(j->ctx->device));
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));  // no launch in flight reads the plan that is about to go
    j->border = border_mode;
    j->plan_ref.reset();
    j->pc_plan = nullptr;
    return P2P_OK;
}

int job_set_view_mask(p2p_job* j, const uint8_t* mask)
{
    if (!j)
        return fail(P2P_ERR_INVALID, "job is NULL");
    const p2p_job_desc& d = j->d;
    HIP_TRY(hipSetDevice(j->ctx->device));
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));  // no launch in flight reads the mask that is about to change
    j->mask_gen++;  // (the pair-context table holds the mask's "not wanted" class)
    if (!mask) {
        (void)dev_free(j->d_view_mask);
        j->d_view_mask = nullptr;
        j->mask_words = 0;
        j->n_views_wanted = d.n_yaw * d.n_pitch;
        return P2P_OK;
    }
    const int words = (d.n_yaw + 31) / 32;
    std::vector<uint32_t> bits((size_t)d.n_pitch * words, 0u);
    int wanted = 0;
    for (int y = 0; y < d.n_yaw; ++y)
        for (int p = 0; p < d.n_pitch; ++p)
            if (mask[(size_t)y * d.n_pitch + p]) {
                bits[(size_t)p * words + (y >> 5)] |= 1u << (y & 31);
                ++wanted;
            }
    if (!j->d_view_mask)
        HIP_TRY(dev_alloc((void**)&j->d_view_mask, bits.size() * sizeof(uint32_t)));
    {
        // on the stream the kernels that read the mask run on, like every other upload of this file (`bits` outlives it)
        StreamSyncGuard sync_on_exit(j->ctx->stream);
        HIP_TRY(hipMemcpyAsync(j->d_view_mask, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice, j->ctx->stream));
        HIP_TRY(hipStreamSynchronize(j->ctx->stream));
        sync_on_exit.armed = false;
    }
    j->mask_words = words;
    j->n_views_wanted = wanted;
    return P2P_OK;
}


#ifdef P2P_AUDIT
// audit build: wait for the launch and read the kernels' violation record
int audit_check(p2p_ctx* ctx, const char* what)
{
    uint32_t rec[p2p::AUDIT_WORDS] = {0};
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(rec, ctx->d_audit, sizeof(rec), hipMemcpyDeviceToHost));
    if (rec[0]) {
        HIP_TRY(hipMemset(ctx->d_audit, 0, sizeof(rec)));
        return fail(P2P_ERR_HIP, "AUDIT %s: site 0x%x block (%u, %u, %u) thread %u value %u limit %u", what, rec[1], rec[2],
                    rec[3], rec[4], rec[5], rec[6], rec[7]);
    }
    return P2P_OK;
}
#endif


int job_run(p2p_job* j)
{
    if (!j)
        return fail(P2P_ERR_INVALID, "job is NULL");
    const p2p_job* so = j->owns_src ? j : j->src_owner;
    for (int i = 0; i < j->d.n_panos; ++i)
        if (!so->pano_set[i])
            return fail(P2P_ERR_STATE, "panorama %d was never set", i);
    HIP_TRY(hipSetDevice(j->ctx->device));
    // behind the uploads into the panoramas it reads and the downloads of the views it is about to overwrite
    // (a finished copy needs no wait any more: one hipEventQuery instead of a barrier packet per launch)
    if (so->up_pending) {
        if (hipEventQuery(so->ev_up) == hipSuccess)
            const_cast<p2p_job*>(so)->up_pending = false;
        else
            HIP_TRY(hipStreamWaitEvent(j->ctx->stream, so->ev_up, 0));
    }
    if (j->down_pending) {
        if (hipEventQuery(j->ev_down) != hipSuccess)
            HIP_TRY(hipStreamWaitEvent(j->ctx->stream, j->ev_down, 0));
        j->down_pending = false;
    }
    p2p::ViewsParams P{};
    P.src = j->d_src;
    P.pano_stride = j->pano_stride;
    P.src_pitch = j->src_pitch;
    P.pw = j->d.pw;
    P.ph = j->d.ph;
    P.ytab = j->d_ytab;
    P.ydesc = j->d_ydesc;
    P.f4tab = j->d_f4tab;
    P.n_yaw = j->d.n_yaw;
    P.n_pitch = j->d.n_pitch;
    P.n_panos = j->d.n_panos;
    P.n_yaw_magic = (uint32_t)(((1ull << 32) + (uint64_t)j->d.n_yaw - 1) / (uint64_t)j->d.n_yaw);
    P.pitch = j->d_pitch;
    P.geom = j->geom;
    P.audit = j->ctx->d_audit;
    P.ow = j->d.ow;
    P.out_row = j->out_row;
    P.view_bytes = (size_t)j->d.oh * j->out_row;
    P.oh = j->d.oh;
    P.out = j->d_out;
    P.border = j->border;
    P.view_mask = j->d_view_mask;
    P.mask_words = j->mask_words;
    const bool timed = j->time_launches && j->ring_pairs > 0;
    const int slot = timed ? (int)(j->runs % j->ring_pairs) : 0;
    const bool float_path = (j->d.flags & (P2P_FLAG_PIXELS_F32 | P2P_FLAG_PIXELS_F16)) != 0;
    if (float_path && j->host_maps)
        return fail(P2P_ERR_STATE, "the float pixel path evaluates its own maps; caller maps are not supported");
    const Options& opt = j->opt;
    // what does not depend on the plan's lists
    auto plan_params = [&](const Plan& Pl) {
        P.coords = Pl.d_coords; P.hdr = Pl.d_hdr; P.px = Pl.d_px; P.items = Pl.d_items; P.px2 = Pl.d_px2;
        P.pairs_per_block = choose_pairs_per_block(j->d, shape_ops(j->shape).shape, opt);
        P.chunk_outer = opt.chunk_outer >= 0 ? opt.chunk_outer : (j->d.n_panos > 1 ? 1 : 0);
        P.main_span = choose_main_span(j->d, shape_ops(j->shape).shape, opt, P.pairs_per_block);
        const int pair_chunks = (j->d.n_panos * j->d.n_yaw + P.pairs_per_block - 1) / P.pairs_per_block;
        P.main_chunks = (pair_chunks + P.main_span - 1) / P.main_span;
        // table-prefetch workgroups (p2p_tile.h: main_block_role) when one launch's plan tables cannot stay in the Infinity
        // Cache next to the panorama (config 4: 565 MB): 2 groups = 64 tiles of lead per XCD
        const size_t table_bytes = plan_table_bytes(j->d, shape_ops(j->shape).shape);
        P.pf_lead = opt.prefetch_lead >= 0 ? opt.prefetch_lead : (table_bytes > ((size_t)128 << 20) ? 2 : 0);
        P.pitch_order = j->d_pitch_order;
        P.main_tail = 0;
        P.main_tail_parts = opt.main_tail_parts;
        P.odd_pairs = j->d_odd_pairs;
        P.n_odd_pairs = j->n_odd_pairs;
        P.rest_ppb = std::min(16, std::max(1, j->n_odd_pairs));
        P.use_pair_list = 0;
    };
    // the main kernel's grid: list order (source bands, all pitch views together) unless the plan has no list
    auto list_params = [&](const Plan& Pl, int main_order) {
        P.main_list = (main_order == 2 || (main_order == 1 && j->d.n_panos == 1)) ? Pl.d_main_list : nullptr;
        P.main_stride = Pl.main_stride;
        P.main_count = Pl.d_main_count;
        const int pair_chunks = (j->d.n_panos * j->d.n_yaw + P.pairs_per_block - 1) / P.pairs_per_block;
        // entries drawn for one chunk of pairs before the next chunk: with ONE panorama a short run (the entries' plan
        // tables and source rows are still in L2 for the next chunk), with several all of them (a chunk's panoramas serve
        // every tile before the next ones are touched: see pair_chunk)
        P.main_group = P.chunk_outer ? std::max(1, Pl.main_stride)
                                     : std::max(1, std::min(Pl.main_stride, choose_main_group(opt, j->shape, P.main_span, pair_chunks)));
        // The last entries of every XCD's list as two workgroups of half the pairs each: when the list runs out, the
        // workgroups in flight end over a whole workgroup's life (25 us on config 2) with ever fewer of them left -- half
        // of that is lost.  Shorter workgroups at the end shorten it.  Only where one workgroup draws ALL pairs of its tile.
        P.main_tail = 0;
        P.main_tail_parts = opt.main_tail_parts;
        if (P.main_list && j->shape != 1 && pair_chunks == 1 && P.main_span == 1 && P.pf_lead == 0 && j->d.n_panos * j->d.n_yaw >= 4) {  // (not the 128-wide kernel: p2p_views.hip)
            const int in_flight = 32 * (j->shape == 1 ? 3 : (j->shape == 2 ? 5 : 7));  // workgroups an XCD holds at a time
            P.main_tail = opt.main_tail >= 0 ? opt.main_tail : in_flight / 5;
            P.main_tail = std::min(P.main_tail, Pl.main_stride);
        }
    };
    // The job's pair-context table (pair_ctx_kernel) for the plan `Pl`: written on `st` when the one the job holds is
    // of another plan, yaw list, view mask or chunking; sets P.pair_ctx.  Not beyond 64 MB.
    auto pair_ctx_table = [&](const Plan* Pl, bool band_plan, hipStream_t st) -> int {
        P.pair_ctx = nullptr;
        P.pair_ctx_chunks = 0;
        const int chunks_all = (j->d.n_panos * j->d.n_yaw + P.pairs_per_block - 1) / P.pairs_per_block;
        const size_t tslots = band_plan ? (size_t)Pl->band_tiles : j->n_tiles * (size_t)j->d.n_pitch;
        const size_t bytes = tslots * (size_t)chunks_all * 64 * sizeof(uint4);
        if (tslots == 0 || bytes > ((size_t)64 << 20) || P.pairs_per_block > 64)
            return P2P_OK;
        const bool stale = !j->d_pair_ctx || j->pc_plan != Pl || j->pc_yaw != j->yaw_ref.get() ||
                           j->pc_mask_gen != j->mask_gen || j->pc_ppb != P.pairs_per_block || j->pc_chunks != chunks_all ||
                           j->pc_slots != tslots;
        if (stale) {
            if (j->d_pair_ctx && (j->pc_slots * (size_t)j->pc_chunks != tslots * (size_t)chunks_all)) {
                HIP_TRY(hipStreamSynchronize(j->ctx->stream));
                (void)dev_free(j->d_pair_ctx);
                j->d_pair_ctx = nullptr;
            }
            if (!j->d_pair_ctx)
                HIP_TRY(dev_alloc((void**)&j->d_pair_ctx, bytes));
            HIP_TRY(shape_ops(j->shape).pair_ctx(P, j->d_pair_ctx, (int)tslots, chunks_all, band_plan ? 1 : 0, st));
            j->pc_plan = Pl; j->pc_yaw = j->yaw_ref.get(); j->pc_mask_gen = j->mask_gen;
            j->pc_ppb = P.pairs_per_block; j->pc_chunks = chunks_all; j->pc_slots = tslots;
        }
        P.pair_ctx = j->d_pair_ctx;
        P.pair_ctx_chunks = chunks_all;
        return P2P_OK;
    };
    bool early_main = false;
    job_settle_shape(j);
    if (j->plan_ref && j->plan_ref->band != job_wants_band(j)) {
        // (the yaws changed under a band plan, or away from one: p2p_job_set_yaws / p2p_job_set_maps)
        HIP_TRY(hipStreamSynchronize(j->ctx->stream));
        j->plan_ref.reset();
    }
    if (!j->plan_ref) {
        // One image through a fresh geometry: the main kernel goes out in grid order right behind the plan pass (it draws
        // the LDS-scheme tiles, whichever they turn out to be); the gather tiles' count, the lists and the other kernels
        // follow below.  The kernels write disjoint pixels, in any order.
        std::function<int(Plan&)> launch_main;
        // (not when P2P_MAIN_ORDER names an order: that launch is the one a test or a tool wants to see)
        if (!float_path && opt.force_rest == 0 && opt.early_main != 0 && opt.scramble_plan == 0 && opt.main_order < 0)
            launch_main = [&](Plan& Pl) -> int {
                plan_params(Pl);
                P.main_list = nullptr;
                P.main_stride = 0;
                P.main_group = 1;
                P.pair_ctx = nullptr;
                P.pair_ctx_chunks = 0;
                // P2P_DEFER_LISTS=0: the per-XCD lists and the pair contexts, which need the plan pass only, go out between
                // the plan pass and the main kernel, and the geometry's first image is drawn as every later one will be.
                // Not the default: 17 + 5 us in front of a kernel that is, on a cold cache, SLOWER in list order (the steady
                // state has it the other way round); Plan::lists_pending has the three orders' times.  The lists on a
                // second stream NEXT to the main kernel (tools/platform/side_stream_probe.hip): the main kernel's seven
                // workgroups per CU hold 504 of a SIMD's 512 registers -- no wave of another kernel starts until they
                // drain, whichever stream was given its work first.
                const int main_order = job_main_order(j);
                if (main_order != 0 && opt.defer_lists == 0) {
                    if (int rc = plan_enqueue_main_lists(Pl, j->n_tiles * (size_t)j->d.n_pitch, shape_ops(j->shape).shape.tile_w, j->ctx->stream))
                        return rc;
                    Pl.lists_made = true;
                    list_params(Pl, main_order);
                    if (opt.pair_ctx_table != 0)
                        if (int rc = pair_ctx_table(&Pl, false, j->ctx->stream))
                            return rc;
                }
                if (timed)
                    HIP_TRY(hipEventRecord(j->ev_ring[2 * slot], j->ctx->stream));
                HIP_TRY(shape_ops(j->shape).views(P, 0, j->ctx->stream));
                early_main = true;
                return P2P_OK;
            };
        int rc = job_build_plan(j, launch_main);
        if (rc != P2P_OK)
            return rc;
    }
    const bool band = j->plan_ref->band;
    if (j->plan_ref->lists_pending && j->plan_ref->launches > 0) {
        int rc = plan_make_main_lists(j, *j->plan_ref);
        if (rc != P2P_OK)
            return rc;
    }
    {   // the job's view of its plan
        const Plan& Pl = *j->plan_ref;
        j->d_coords = Pl.d_coords; j->d_hdr = Pl.d_hdr; j->d_px = Pl.d_px; j->d_items = Pl.d_items; j->d_px2 = Pl.d_px2;
        j->d_gather_list = Pl.d_gather_list; j->d_xcd_list = Pl.d_xcd_list; j->d_xcd_all = Pl.d_xcd_all;
        j->xcd_stride = Pl.xcd_stride; j->xcd_all_stride = Pl.xcd_all_stride; j->n_gather = Pl.n_gather;
        j->d_main_list = Pl.d_main_list; j->main_stride = Pl.main_stride; j->d_main_count = Pl.d_main_count;
    }
    P.pairs_per_block = choose_pairs_per_block(j->d, shape_ops(j->shape).shape, opt);
    P.chunk_outer = opt.chunk_outer >= 0 ? opt.chunk_outer : (j->d.n_panos > 1 ? 1 : 0);
    P.main_span = choose_main_span(j->d, shape_ops(j->shape).shape, opt, P.pairs_per_block);
    const int pair_chunks = (j->d.n_panos * j->d.n_yaw + P.pairs_per_block - 1) / P.pairs_per_block;
    P.main_chunks = (pair_chunks + P.main_span - 1) / P.main_span;
    // table-prefetch workgroups (p2p_tile.h: main_block_role) when one launch's plan tables cannot stay in the Infinity
    // Cache next to the panorama (config 4: 565 MB): 2 groups = 64 tiles of lead per XCD
    {
        const size_t table_bytes = plan_table_bytes(j->d, shape_ops(j->shape).shape);
        P.pf_lead = opt.prefetch_lead >= 0 ? opt.prefetch_lead : (table_bytes > ((size_t)128 << 20) ? 2 : 0);
    }
    list_params(*j->plan_ref, band ? 0 : job_main_order(j));
    if (band) {
        const Plan& Pl = *j->plan_ref;
        P.band_hdr = Pl.d_band_hdr; P.band_px = Pl.d_band_px; P.band_grp = Pl.d_band_grp; P.band_info = Pl.d_band_info;
        P.band_tiles = Pl.band_tiles;
        P.band_per = Pl.band_per;
        P.pf_lead = 0;
        // the split tail (see main_tail): one chunk of pairs, no span loop
        P.band_tail = 0;
        if (pair_chunks == 1 && P.main_span == 1 && j->d.n_panos * j->d.n_yaw >= 4) {
            const int in_flight = 32 * (j->shape == 1 ? 3 : (j->shape == 2 ? 5 : 7));
            P.band_tail = opt.main_tail >= 0 ? opt.main_tail : in_flight / 5;
            P.band_tail = std::min(P.band_tail, std::max(0, Pl.band_tiles / 8));
        }
    }
    P.pitch_order = j->d_pitch_order;
    P.coords = j->d_coords;
    P.hdr = j->d_hdr;
    P.px = j->d_px;
    P.items = j->d_items;
    P.gather_list = j->d_gather_list;
    P.n_gather = j->n_gather;
    // With view rows of whole dwords the main and the gather kernel draw every plain-shift yaw, and the rest / table
    // kernels only the listed odd pairs (up to 16 per workgroup: one set-up for all of them); otherwise those two draw all
    const bool fast_width = opt.force_rest == 0;  // (diagnosis: 1 = everything through the general loops)
    const bool gather_ok = fast_width && j->border == 0;  // the gather kernel: BORDER_CONSTANT 0
    P.odd_pairs = j->d_odd_pairs;
    P.n_odd_pairs = j->n_odd_pairs;
    P.rest_ppb = std::min(16, std::max(1, j->n_odd_pairs));
    {
        // pairs per workgroup of the gather / table kernels: about 4096 workgroups in all, at most 16 pairs each
        // (the tile's coordinates and weights are set up once per workgroup)
        const long long np = (long long)j->d.n_panos * j->d.n_yaw;
        long long ppb = (np * std::max(1, j->n_gather) + 2047) / 2048;
        const long long cap = opt.gather_ppb;
        P.gather_ppb = (int)std::min<long long>(std::max<long long>(ppb, 1), std::min<long long>(np, std::min<long long>(cap, 64)));
    }
    if (float_path) {
        // opt-in float pixel path (beyond the reference): one float resample per view, see p2p_float.hip
        P.px2 = j->d_px2;
        P.yaw_rad = j->d_yaw_rad;
        const bool half = (j->d.flags & P2P_FLAG_PIXELS_F16) != 0;
        if (timed)
            HIP_TRY(hipEventRecord(j->ev_ring[2 * slot], j->ctx->stream));
        if (j->n_gather > 0)
            HIP_TRY(shape_ops(j->shape).float_views(P, half, 1, j->ctx->stream));
        HIP_TRY(shape_ops(j->shape).float_views(P, half, 0, j->ctx->stream));
        if (timed)
            HIP_TRY(hipEventRecord(j->ev_ring[2 * slot + 1], j->ctx->stream));
#ifdef P2P_AUDIT
        if (int rc = audit_check(j->ctx, "float views"))
            return rc;
#endif
        j->run_unmarked = true;
        if (!j->owns_src)
            j->src_owner->run_unmarked = true;
        j->runs++;
        j->ran = true;
        return P2P_OK;
    }
    // the pair-context table: not FOR a job's very first launch (the main kernel is already out), but made behind it
    // when the plan's lists were (P2P_DEFER_LISTS=1)
    P.pair_ctx = nullptr;
    P.pair_ctx_chunks = 0;
    if (opt.pair_ctx_table != 0 && opt.force_rest == 0 && (!early_main || (opt.defer_lists == 1 && j->plan_ref->lists_made && !band)))
        if (int rc = pair_ctx_table(j->plan_ref.get(), band, j->ctx->stream))
            return rc;
    if (early_main) {
        P.pair_ctx = nullptr;
        P.pair_ctx_chunks = 0;
    }
    if (timed && !early_main)
        HIP_TRY(hipEventRecord(j->ev_ring[2 * slot], j->ctx->stream));
    // The main kernel draws every LDS-scheme tile for every yaw that is a plain shift, the gather kernel every other
    // tile for those yaws -- on the reference's own workloads that is everything.  The other two kernels are launched
    // only when the yaw tables or the job's shape call for them; the four write disjoint pixels.
    const size_t slots = j->n_tiles * (size_t)j->d.n_pitch;
    const bool any_lds = (size_t)j->n_gather < slots;  // a tile the LDS-scheme kernels draw
    const bool need_rest = any_lds && (j->n_odd_yaws > 0 || !fast_width);
    // Few tiles left for the LDS scheme (the edge tiles of a strongly minifying view set): the gather kernel, which
    // needs nothing but the coordinates, draws those too, and the main kernel's launch (6 us for a handful of
    // tiles) is saved.  The odd pairs of those tiles stay the rest kernel's.
    P.gather_all = (gather_ok && !band && !early_main && j->n_gather > 0 && j->d_xcd_all && (slots - (size_t)j->n_gather) * 4 <= slots &&
                    opt.gather_all != 0) ? 1 : 0;
    // (the gather kernel of a big job on a side stream, forked and joined by events, so that its cache waits overlap
    // the main kernel's arithmetic: config 4's pitch 30 1647 vs 1621 us, all five pitches 8021 vs 7988 -- the two
    // kernels do not interleave, not kept.  Round 4 once more, the side stream at the LOWEST priority and the gather
    // kernel enqueued behind the main kernel, to fill the slots its last workgroups leave: config 4 6.301 / 6.312 /
    // 6.303 -> 6.286 / 6.311 / 6.293 ms, five 1080p pitch views x 12 yaws 170.6 -> 174.2 us: not kept either)
    if (P.gather_all && !j->plan_ref->coords_full)
        if (int rc = ensure_full_coords(j))
            return rc;
    // Band plans: the gather kernel's few, long workgroups (the tiles around a pole) are a chain of latencies -- 17 us as a
    // launch of their own, with the band kernel behind them waiting for the last one; they become the first workgroups of
    // the band kernel's own launch (with one set of tap registers: the kernel keeps six waves per SIMD).
    // ... and of the main kernel's, in list order (a one-dimensional grid; not on a job's first launch, which is in grid
    // order, and not where the gather kernel draws every tile anyway).
    const bool merged = j->n_gather > 0 && gather_ok && opt.merge_gather != 0 && j->n_odd_pairs == 0 &&
                        (band ? j->plan_ref->band_tiles > 0
                              : (P.main_list != nullptr && !early_main && !P.gather_all && any_lds && fast_width));
    if (j->n_gather > 0) {
        if (gather_ok) {
            P.use_pair_list = 0;
            if (P.gather_all) {
                const long long np = (long long)j->d.n_panos * j->d.n_yaw;
                const long long ppb = (np * (long long)slots + 2047) / 2048;
                P.gather_ppb = (int)std::min<long long>(std::max<long long>(ppb, 1), std::min<long long>(np, std::min(16, std::max(1, opt.gather_ppb))));
            }
            P.gather_list = P.gather_all ? j->d_xcd_all : j->d_xcd_list;
            P.n_list = P.gather_all ? j->xcd_all_stride : j->xcd_stride;
            if (P.gather_list && P.n_list > 0 && !merged)
                HIP_TRY(shape_ops(j->shape).views(P, 3, j->ctx->stream));
            if (merged) {
                P.merge_gather_list = P.gather_list;
                P.merge_gather_n = P.n_list;
            }
            P.gather_list = j->d_gather_list;
        }
        // the table kernel: every pair where the gather kernel does not apply, else the odd pairs
        P.use_pair_list = gather_ok ? 1 : 0;
        if (!gather_ok || j->n_odd_pairs > 0)
            HIP_TRY(shape_ops(j->shape).views(P, 2, j->ctx->stream));
    }
    if (need_rest && !band) {
        // (a yaw row that is not a shift is gathered per pixel from the coordinates, also on the LDS-scheme tiles)
        bool not_a_shift = !fast_width;
        if (j->yaw_ref)
            for (const auto& yd : j->yaw_ref->desc)
                not_a_shift = not_a_shift || yd.mode == 2;
        if (not_a_shift && !j->plan_ref->coords_full)
            if (int rc = ensure_full_coords(j))
                return rc;
        P.use_pair_list = (fast_width && j->n_odd_pairs > 0) ? 1 : 0;
        HIP_TRY(shape_ops(j->shape).views(P, 1, j->ctx->stream));
    }
    P.use_pair_list = 0;
    if (band) {
        if (P.band_tiles > 0)
            HIP_TRY(shape_ops(j->shape).views(P, 4, j->ctx->stream));
    } else if (fast_width && any_lds && !P.gather_all && !early_main)
        HIP_TRY(shape_ops(j->shape).views(P, 0, j->ctx->stream));
    if (timed)
        HIP_TRY(hipEventRecord(j->ev_ring[2 * slot + 1], j->ctx->stream));
#ifdef P2P_AUDIT
    if (int rc = audit_check(j->ctx, "views"))
        return rc;
#endif
    // (the event that orders copies behind this run is recorded when a copy asks for it: mark_run)
    j->plan_ref->launches++;
    j->run_unmarked = true;
    if (!j->owns_src)
        j->src_owner->run_unmarked = true;  // uploads into the shared panoramas wait for this run too
    j->runs++;
    j->ran = true;
    return P2P_OK;
}

int job_plan_ms(p2p_job* j, float* plan_ms, float* tables_ms)
{
    if (!j || !plan_ms || !tables_ms)
        return fail(P2P_ERR_INVALID, "NULL argument");
    if (!j->plan_ref || !j->yaw_ref)
        return fail(P2P_ERR_STATE, "p2p_job_run has not been called");
    *plan_ms = j->plan_ref->plan_ms;
    *tables_ms = j->yaw_ref->tables_ms;
    return P2P_OK;
}

int job_time_launches(p2p_job* j, int n)
{
    if (!j)
        return fail(P2P_ERR_INVALID, "job is NULL");
    if (n < 0 || n > kEvRingMax)
        return fail(P2P_ERR_INVALID, "p2p_job_time_launches: n must be 0 (off) .. %d launches to keep", kEvRingMax);
    HIP_TRY(hipSetDevice(j->ctx->device));
    if (n > j->ring_pairs) {  // the ring grows to what was asked for and is kept (two events per launch to keep)
        HIP_TRY(hipStreamSynchronize(j->ctx->stream));
        j->ev_ring.reserve(2 * (size_t)n);
        while ((int)j->ev_ring.size() < 2 * n) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            j->ev_ring.push_back(e);
        }
        j->ring_pairs = n;
    }
    j->time_launches = n != 0;
    j->runs = 0;  // the ring only describes launches made in the current mode
    return P2P_OK;
}

int ctx_mark(p2p_ctx* c, int which)
{
    if (!c || (which != 0 && which != 1))
        return fail(P2P_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(which ? c->ev1 : c->ev0, c->stream));
    return P2P_OK;
}

int ctx_marked_ms(p2p_ctx* c, float* ms)
{
    if (!c || !ms)
        return fail(P2P_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return P2P_OK;
}

int job_kernel_ms(p2p_job* j, float* ms)
{
    if (!j || !ms)
        return fail(P2P_ERR_INVALID, "NULL argument");
    if (!j->ran || !j->time_launches || j->runs < 1 || j->ring_pairs < 1)
        return fail(P2P_ERR_STATE, "no timed p2p_job_run has been made (job_time_launches(job, n) turns the timing on)");
    HIP_TRY(hipSetDevice(j->ctx->device));
    const int slot = (int)((j->runs - 1) % j->ring_pairs);
    HIP_TRY(hipEventSynchronize(j->ev_ring[2 * slot + 1]));
    HIP_TRY(hipEventElapsedTime(ms, j->ev_ring[2 * slot], j->ev_ring[2 * slot + 1]));
    return P2P_OK;
}

int job_kernel_ms_last(p2p_job* j, float* ms, int n)
{
    if (!j || !ms || n < 1)
        return fail(P2P_ERR_INVALID, "bad argument");
    if (!j->time_launches || j->runs < n || n > j->ring_pairs)
        return fail(P2P_ERR_STATE, "only %lld timed runs recorded (the ring holds %d: p2p_job_time_launches)", j->time_launches ? j->runs : 0LL, j->ring_pairs);
    HIP_TRY(hipSetDevice(j->ctx->device));
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));
    for (int k = 0; k < n; ++k) {
        const int slot = (int)((j->runs - n + k) % j->ring_pairs);
        HIP_TRY(hipEventElapsedTime(&ms[k], j->ev_ring[2 * slot], j->ev_ring[2 * slot + 1]));
    }
    return P2P_OK;
}

// the views of panorama `index` -> the caller's contiguous [n_yaw][n_pitch][oh][ow][3] array, on `st`
int enqueue_views_copy(p2p_job* j, int index, uint8_t* out, hipStream_t st)
{
    const size_t per = j->out_bytes / j->d.n_panos;
    const size_t row = (size_t)3 * j->d.ow;
    if ((size_t)j->out_row == row) {
        HIP_TRY(hipMemcpyAsync(out, j->d_out + per * index, per, hipMemcpyDeviceToHost, st));
    } else {
        // device rows are padded to whole 4-pixel groups, the caller's array is not: packed on the device (a row-wise
        // DMA copy costs microseconds per row), then one copy
        const size_t rows = (size_t)j->d.n_yaw * j->d.n_pitch * j->d.oh, packed = rows * row;
        if (!j->d_pack)
            HIP_TRY(dev_alloc((void**)&j->d_pack, (packed + 3) & ~(size_t)3));
        HIP_TRY(p2p::launch_compact_rows(j->d_pack, j->d_out + per * index, packed, (int)row, j->out_row, st));
        HIP_TRY(hipMemcpyAsync(out, j->d_pack, packed, hipMemcpyDeviceToHost, st));
    }
    return P2P_OK;
}

// ONE view (panorama `index`, yaw yaw_i, pitch pitch_i) -> the caller's contiguous [oh][ow][3] array, on `st`
int enqueue_view_copy(p2p_job* j, int index, int yaw_i, int pitch_i, uint8_t* out, hipStream_t st)
{
    const size_t view = (size_t)j->d.oh * j->out_row;
    const uint8_t* src = j->d_out + (((size_t)index * j->d.n_yaw + yaw_i) * j->d.n_pitch + pitch_i) * view;
    const size_t row = (size_t)3 * j->d.ow;
    if ((size_t)j->out_row == row) {
        HIP_TRY(hipMemcpyAsync(out, src, view, hipMemcpyDeviceToHost, st));
    } else {
        const size_t packed = (size_t)j->d.oh * row;
        if (!j->d_pack) {  // sized for a whole panorama's views, as the whole-block download uses it
            const size_t all = (size_t)j->d.n_yaw * j->d.n_pitch * packed;
            HIP_TRY(dev_alloc((void**)&j->d_pack, (all + 3) & ~(size_t)3));
        }
        HIP_TRY(p2p::launch_compact_rows(j->d_pack, src, packed, (int)row, j->out_row, st));
        HIP_TRY(hipMemcpyAsync(out, j->d_pack, packed, hipMemcpyDeviceToHost, st));
    }
    return P2P_OK;
}

int get_views_check(p2p_job* j, int index, uint8_t* out)
{
    if (!j || !out)
        return fail(P2P_ERR_INVALID, "NULL argument");
    if (index < 0 || index >= j->d.n_panos)
        return fail(P2P_ERR_INVALID, "panorama index %d out of range", index);
    if (!j->ran)
        return fail(P2P_ERR_STATE, "p2p_job_run has not been called");
    return P2P_OK;
}

int job_get_views_async(p2p_job* j, int index, uint8_t* out)
{
    if (int rc = get_views_check(j, index, out))
        return rc;
    HIP_TRY(hipSetDevice(j->ctx->device));
    hipStream_t down = nullptr;
    HIP_TRY(ctx_copy_stream(j->ctx, false, &down));
    // on the download stream, behind the job's last run: the copy overlaps other jobs' kernels and uploads
    if (int rc = mark_run(j))
        return rc;
    HIP_TRY(hipStreamWaitEvent(down, j->ev_run, 0));
    if (int rc = enqueue_views_copy(j, index, out, down))
        return rc;
    HIP_TRY(hipEventRecord(j->ev_down, down));
    j->down_pending = true;
    return P2P_OK;
}

int job_get_view_async(p2p_job* j, int index, int yaw_i, int pitch_i, uint8_t* out)
{
    if (int rc = get_views_check(j, index, out))
        return rc;
    if (yaw_i < 0 || yaw_i >= j->d.n_yaw || pitch_i < 0 || pitch_i >= j->d.n_pitch)
        return fail(P2P_ERR_INVALID, "view (yaw %d, pitch %d) out of range", yaw_i, pitch_i);
    if ((size_t)j->out_row != (size_t)3 * j->d.ow)
        return fail(P2P_ERR_STATE, "asynchronous single-view downloads need a view width divisible by 4 (the packing buffer is shared); use p2p_job_get_view");
    HIP_TRY(hipSetDevice(j->ctx->device));
    hipStream_t down = nullptr;
    HIP_TRY(ctx_copy_stream(j->ctx, false, &down));
    if (int rc = mark_run(j))
        return rc;
    HIP_TRY(hipStreamWaitEvent(down, j->ev_run, 0));
    if (int rc = enqueue_view_copy(j, index, yaw_i, pitch_i, out, down))
        return rc;
    HIP_TRY(hipEventRecord(j->ev_down, down));
    j->down_pending = true;
    return P2P_OK;
}

// rows [row0, row1) of one view, packed (3 * ow bytes per row), behind the last run -- the download that goes with
// p2p_job_set_rows.  The asynchronous form needs a view width divisible by 4, like p2p_job_get_view_async.
int enqueue_rows_copy(p2p_job* j, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out, hipStream_t st)
{
    const size_t view = (size_t)j->d.oh * j->out_row;
    const uint8_t* src = j->d_out + (((size_t)index * j->d.n_yaw + yaw_i) * j->d.n_pitch + pitch_i) * view + (size_t)row0 * j->out_row;
    const size_t row = (size_t)3 * j->d.ow, rows = (size_t)(row1 - row0);
    if ((size_t)j->out_row == row) {
        HIP_TRY(hipMemcpyAsync(out, src, rows * row, hipMemcpyDeviceToHost, st));
    } else {
        if (!j->d_pack) {  // sized for a whole panorama's views, as the whole-block download uses it
            const size_t all = (size_t)j->d.n_yaw * j->d.n_pitch * j->d.oh * row;
            HIP_TRY(dev_alloc((void**)&j->d_pack, (all + 3) & ~(size_t)3));
        }
        HIP_TRY(p2p::launch_compact_rows(j->d_pack, src, rows * row, (int)row, j->out_row, st));
        HIP_TRY(hipMemcpyAsync(out, j->d_pack, rows * row, hipMemcpyDeviceToHost, st));
    }
    return P2P_OK;
}

int view_rows_check(p2p_job* j, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out)
{
    if (int rc = get_views_check(j, index, out))
        return rc;
    if (yaw_i < 0 || yaw_i >= j->d.n_yaw || pitch_i < 0 || pitch_i >= j->d.n_pitch)
        return fail(P2P_ERR_INVALID, "view (yaw %d, pitch %d) out of range", yaw_i, pitch_i);
    if (row0 < 0 || row1 <= row0 || row1 > j->d.oh)
        return fail(P2P_ERR_INVALID, "rows [%d, %d) of %d", row0, row1, j->d.oh);
    return P2P_OK;
}

int job_get_view_rows_async(p2p_job* j, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out)
{
    if (int rc = view_rows_check(j, index, yaw_i, pitch_i, row0, row1, out))
        return rc;
    if ((size_t)j->out_row != (size_t)3 * j->d.ow)
        return fail(P2P_ERR_STATE, "asynchronous row downloads need a view width divisible by 4 (the packing buffer is shared); use p2p_job_get_view_rows");
    HIP_TRY(hipSetDevice(j->ctx->device));
    hipStream_t down = nullptr;
    HIP_TRY(ctx_copy_stream(j->ctx, false, &down));
    if (int rc = mark_run(j))
        return rc;
    HIP_TRY(hipStreamWaitEvent(down, j->ev_run, 0));
    if (int rc = enqueue_rows_copy(j, index, yaw_i, pitch_i, row0, row1, out, down))
        return rc;
    HIP_TRY(hipEventRecord(j->ev_down, down));
    j->down_pending = true;
    return P2P_OK;
}

int job_get_view_rows(p2p_job* j, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out)
{
    if (int rc = view_rows_check(j, index, yaw_i, pitch_i, row0, row1, out))
        return rc;
    HIP_TRY(hipSetDevice(j->ctx->device));
    if (j->down_pending) {  // d_pack is shared with the asynchronous path
        HIP_TRY(hipEventSynchronize(j->ev_down));
        j->down_pending = false;
    }
    StreamSyncGuard sync_on_exit(j->ctx->stream);  // `out` is the caller's
    if (int rc = enqueue_rows_copy(j, index, yaw_i, pitch_i, row0, row1, out, j->ctx->stream))
        return rc;
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));
    sync_on_exit.armed = false;
    return P2P_OK;
}

int job_get_view(p2p_job* j, int index, int yaw_i, int pitch_i, uint8_t* out)
{
    if (int rc = get_views_check(j, index, out))
        return rc;
    if (yaw_i < 0 || yaw_i >= j->d.n_yaw || pitch_i < 0 || pitch_i >= j->d.n_pitch)
        return fail(P2P_ERR_INVALID, "view (yaw %d, pitch %d) out of range", yaw_i, pitch_i);
    HIP_TRY(hipSetDevice(j->ctx->device));
    if (j->down_pending) {  // d_pack is shared with the asynchronous path
        HIP_TRY(hipEventSynchronize(j->ev_down));
        j->down_pending = false;
    }
    StreamSyncGuard sync_on_exit(j->ctx->stream);  // `out` is the caller's
    if (int rc = enqueue_view_copy(j, index, yaw_i, pitch_i, out, j->ctx->stream))
        return rc;
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));
    sync_on_exit.armed = false;
    return P2P_OK;
}

int job_wait(p2p_job* j)
{
    if (!j)
        return fail(P2P_ERR_INVALID, "job is NULL");
    HIP_TRY(hipSetDevice(j->ctx->device));
    if (j->up_pending)
        HIP_TRY(hipEventSynchronize(j->ev_up));
    if (int rc = mark_run(j))
        return rc;
    if (j->ev_run_recorded)
        HIP_TRY(hipEventSynchronize(j->ev_run));
    if (j->down_pending)
        HIP_TRY(hipEventSynchronize(j->ev_down));
    j->up_pending = false;
    j->down_pending = false;
    return P2P_OK;
}

int job_get_views(p2p_job* j, int index, uint8_t* out)
{
    if (int rc = get_views_check(j, index, out))
        return rc;
    HIP_TRY(hipSetDevice(j->ctx->device));
    // in order on the kernel stream, behind the job's last run (d_pack is shared with the asynchronous path: an
    // asynchronous download still in flight is waited for first)
    if (j->down_pending) {
        HIP_TRY(hipEventSynchronize(j->ev_down));
        j->down_pending = false;
    }
    StreamSyncGuard sync_on_exit(j->ctx->stream);  // `out` is the caller's
    if (int rc = enqueue_views_copy(j, index, out, j->ctx->stream))
        return rc;
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));
    sync_on_exit.armed = false;
    return P2P_OK;
}

void* job_device_out(p2p_job* j, int64_t* bytes)
{
    if (!j)
        return nullptr;
    if (bytes)
        *bytes = (int64_t)j->out_bytes;
    return j->d_out;
}

int job_get_coords(p2p_job* j, int32_t* sxsy)
{
    if (!j || !sxsy)
        return fail(P2P_ERR_INVALID, "NULL argument");
    if (!j->d_coords)
        return fail(P2P_ERR_STATE, "the float pixel paths keep no quantised coordinates");
    if (!j->ran)
        return fail(P2P_ERR_STATE, "p2p_job_run has not been called");
    if (!j->plan_ref)  // (p2p_job_set_maps / p2p_job_set_rows since: the coordinates' plan is gone, the next run makes another)
        return fail(P2P_ERR_STATE, "the job's maps or rows changed since its last run: run it again first");
    HIP_TRY(hipSetDevice(j->ctx->device));
    if (j->plan_ref && !j->plan_ref->coords_full)
        if (int rc = ensure_full_coords(j))
            return rc;
    const size_t n = (size_t)j->d.n_pitch * j->d.oh * j->d.ow * 2 * sizeof(int32_t);
    HIP_TRY(hipMemcpyAsync(sxsy, (const void*)j->d_coords, n, hipMemcpyDeviceToHost, j->ctx->stream));
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));
    return P2P_OK;
}

int job_get_yaw_tables(p2p_job* j, uint32_t* packed)
{
    if (!j || !packed)
        return fail(P2P_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(j->ctx->device));
    const size_t n = (size_t)j->d.n_yaw * j->d.pw * sizeof(uint32_t);
    HIP_TRY(hipMemcpyAsync(packed, j->d_ytab, n, hipMemcpyDeviceToHost, j->ctx->stream));
    HIP_TRY(hipStreamSynchronize(j->ctx->stream));
    return P2P_OK;
}


int job_get_info(p2p_job* j, p2p_job_info* out)
{
    if (!j || !out)
        return fail(P2P_ERR_INVALID, "NULL argument");
    memset(out, 0, sizeof(*out));
    job_settle_shape(j);
    const p2p::TileShape& S = shape_ops(j->shape).shape;
    out->tile_w = S.tile_w;
    out->tile_h = S.tile_h;
    out->n_tiles = (int64_t)j->n_tiles * j->d.n_pitch;
    out->pairs_per_block = choose_pairs_per_block(j->d, S, j->opt);
    out->pair_chunks = (j->d.n_panos * j->d.n_yaw + out->pairs_per_block - 1) / out->pairs_per_block;
    const int mo = job_main_order(j);
    out->list_order = (mo == 2 || (mo == 1 && j->d.n_panos == 1)) ? 1 : 0;

    out->prefetch_lead = j->opt.prefetch_lead >= 0 ? j->opt.prefetch_lead : (plan_table_bytes(j->d, S) > ((size_t)128 << 20) ? 2 : 0);
    out->chunks_per_workgroup = choose_main_span(j->d, S, j->opt, out->pairs_per_block);
    out->main_group = choose_main_group(j->opt, j->shape, out->chunks_per_workgroup, out->pair_chunks);
    out->n_gather_tiles = j->plan_ref ? (int64_t)j->plan_ref->n_gather : -1;
    out->n_odd_yaws = j->n_odd_yaws;
    out->n_views_wanted = j->n_views_wanted;
    out->timing_events = (int32_t)j->ev_ring.size();
    out->copy_streams = (j->ctx->stream_up != nullptr) + (j->ctx->stream_down != nullptr);
    out->band_tiles = j->plan_ref ? (j->plan_ref->band ? j->plan_ref->band_tiles : 0) : (job_wants_band(j) ? -1 : 0);
    out->lds_items_cap = S.cap;
    return P2P_OK;
}

}  // namespace p2p_host
