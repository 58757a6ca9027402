// Drives the C ABI of include/p2p_hip.h through valid and invalid call sequences in the host-only build
// (stub HIP runtime, stub launchers) under -fsanitize=address,undefined: every host-side allocation, copy size,
// index computation and teardown path of 360-to-planer-images_amd/csrc/p2p_abi.cpp + p2p_host_*.cpp, from several threads.
#include "p2p_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <thread>
#include <vector>

// ---- an operator new that fails on request (the exception barrier of the C ABI) ----
// While a thread has armed it, the N-th allocation that thread makes throws std::bad_alloc -- inside whatever
// std::vector / std::map / std::shared_ptr / std::function of the library (or of the stand-in launchers) asked for it.
static thread_local long g_fail_at = -1, g_news = 0;
static thread_local bool g_in_library = false;   // (the harness's own buffers are not the library's allocations)
static void* counted_alloc(size_t n)
{
    if (g_in_library && g_fail_at >= 0 && ++g_news == g_fail_at)
        throw std::bad_alloc();
    void* p = malloc(n ? n : 1);
    if (!p)
        throw std::bad_alloc();
    return p;
}
void* operator new(size_t n) { return counted_alloc(n); }
void* operator new[](size_t n) { return counted_alloc(n); }
void* operator new(size_t n, const std::nothrow_t&) noexcept { try { return counted_alloc(n); } catch (...) { return nullptr; } }
void* operator new[](size_t n, const std::nothrow_t&) noexcept { try { return counted_alloc(n); } catch (...) { return nullptr; } }
void operator delete(void* p, const std::nothrow_t&) noexcept { free(p); }
void operator delete[](void* p, const std::nothrow_t&) noexcept { free(p); }
void operator delete(void* p) noexcept { free(p); }
void operator delete[](void* p) noexcept { free(p); }
void operator delete(void* p, size_t) noexcept { free(p); }
void operator delete[](void* p, size_t) noexcept { free(p); }

extern "C" int p2p_stub_device_count;
extern "C" int p2p_stub_drop_count;
extern "C" long p2p_stub_live(int what);  // 0: events, 1: streams the library holds right now

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed line %d: %s (%s)\n", __LINE__, #cond, p2p_last_error()); exit(1); } } while (0)

static void one_shot_calls(int seed)
{
    const int pw = 64 + 4 * (seed % 3), ph = 32, ow = 70 + seed, oh = 33;
    std::vector<uint8_t> pano((size_t)pw * ph * 3, (uint8_t)seed);
    const int32_t yaws[3] = {0, 30, -400}, pitches[2] = {60, 120};
    std::vector<uint8_t> out((size_t)3 * 2 * oh * ow * 3);
    for (int rep = 0; rep < 3; ++rep)  // second and third call take the cached-job path, with changed yaws
        CHECK(p2p_remap_views_u8(pano.data(), pw, ph, 3 * pw, yaws + (rep == 2), 3 - (rep == 2), pitches, 2, 90, ow, oh,
                                 out.data(), 0, 0) == P2P_OK);
    const double yd[2] = {30.5, 0.25}, pd[3] = {44.5, 0.0, 180.0};
    std::vector<uint8_t> out2((size_t)2 * 3 * oh * ow * 3);
    CHECK(p2p_remap_views_f64(pano.data(), pw, ph, 3 * pw, yd, 2, pd, 3, 72.5, ow, oh, out2.data(), 0, P2P_FLAG_PIXELS_F16) == P2P_OK);
    std::vector<float> rows((size_t)pw, 1.0f), U((size_t)ow * oh, 2.0f), V((size_t)ow * oh, 3.0f);
    std::vector<uint8_t> out3((size_t)oh * ow * 3);
    CHECK(p2p_remap_views_maps_u8(pano.data(), pw, ph, 3 * pw, rows.data(), 1, U.data(), V.data(), 1, ow, oh, out3.data(), 0) == P2P_OK);
    for (int cn : {1, 3, 4})
        for (int inter : {P2P_INTER_NEAREST, P2P_INTER_LINEAR, P2P_INTER_CUBIC}) {
            std::vector<uint8_t> src((size_t)pw * ph * cn, 7), dst((size_t)ow * oh * cn);
            CHECK(p2p_remap_maps_interp_u8(src.data(), pw, ph, pw * cn, cn, U.data(), V.data(), ow, oh, dst.data(), inter,
                                           P2P_BORDER_REFLECT, nullptr, 0) == P2P_OK);
        }
    std::vector<float> mu((size_t)ow * oh), mv((size_t)ow * oh), row(pw);
    CHECK(p2p_build_pitch_map(ow, oh, 1.5, 1.0, pw, ph, mu.data(), mv.data(), 0) == P2P_OK);
    const float R9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    CHECK(p2p_build_rot_map(ow, oh, 1.5, R9, pw, ph, mu.data(), mv.data(), 0) == P2P_OK);
    CHECK(p2p_build_yaw_row(pw, 0.5, row.data(), 0) == P2P_OK);
}

// One job's life through the resident API; returns at the first call that fails (its status in *rc) after tearing down
// what exists.  Every status must be P2P_OK or -- with an allocation failure armed -- P2P_ERR_OOM / P2P_ERR_HIP with a
// message; nothing may abort, leak or leave the library unusable.
static bool job_life(bool band, int* rc_out)
{
    const int pw = 64, ph = 32, ow = 70, oh = 33;
    std::vector<uint8_t> pano((size_t)pw * ph * 3, 5), views((size_t)2 * 2 * oh * ow * 3), one((size_t)oh * ow * 3);
    std::vector<float> U((size_t)2 * oh * ow, 3.0f), V((size_t)2 * oh * ow, 4.0f);
    const int32_t yaws[2] = {0, 90}, pitches[2] = {60, 120};
    const uint8_t mask[4] = {1, 0, 0, 1};
    p2p_ctx* ctx = nullptr;
    p2p_job* job = nullptr;
    p2p_job_desc d = {pw, ph, 1, 2, yaws, 2, pitches, 90, ow, oh, P2P_FLAG_DEFAULT};
    g_in_library = true;
    int rc = p2p_ctx_create(0, &ctx);
    if (rc == P2P_OK) rc = p2p_job_create(ctx, &d, &job);
    if (rc == P2P_OK) rc = p2p_job_set_pano(job, 0, pano.data(), 3 * pw);
    if (rc == P2P_OK) rc = p2p_job_run(job);          // the plan (a band plan with its two-stage build when `band`)
    if (rc == P2P_OK) rc = p2p_job_run(job);          // the deferred per-XCD lists, the pair-context table
    if (rc == P2P_OK) rc = p2p_job_get_views(job, 0, views.data());
    if (rc == P2P_OK) rc = p2p_job_set_view_mask(job, mask);
    if (rc == P2P_OK) rc = p2p_job_run(job);
    if (rc == P2P_OK) rc = p2p_job_get_view(job, 0, 1, 1, one.data());
    if (rc == P2P_OK) rc = p2p_job_set_view_mask(job, nullptr);
    if (rc == P2P_OK && !band) rc = p2p_job_set_maps(job, nullptr, U.data(), V.data());   // a private plan from caller maps
    if (rc == P2P_OK) rc = p2p_job_run(job);
    if (rc == P2P_OK) rc = p2p_job_time_launches(job, 4);
    if (rc == P2P_OK) rc = p2p_job_run(job);
    if (rc == P2P_OK) rc = p2p_job_get_views_async(job, 0, views.data());
    if (rc == P2P_OK) rc = p2p_job_wait(job);
    g_in_library = false;
    if (rc != P2P_OK && !*p2p_last_error()) { fprintf(stderr, "a failing call left no message\n"); exit(1); }
    g_in_library = true;
    p2p_job_destroy(job);
    p2p_ctx_destroy(ctx);
    g_in_library = false;
    *rc_out = rc;
    return rc == P2P_OK;
}

static bool oneshot_life(int* rc_out)
{
    const int pw = 68, ph = 32, ow = 70, oh = 33;
    std::vector<uint8_t> pano((size_t)pw * ph * 3, 5), out((size_t)2 * 2 * oh * ow * 3), dst((size_t)ow * oh * 4);
    std::vector<float> U((size_t)2 * oh * ow, 3.0f), V((size_t)2 * oh * ow, 4.0f);
    const int32_t yaws[2] = {0, 90}, pitches[2] = {60, 120};
    const double yd[2] = {10.0, 20.0};
    g_in_library = true;
    int rc = p2p_remap_views_u8(pano.data(), pw, ph, 3 * pw, yaws, 2, pitches, 2, 90, ow, oh, out.data(), 0, 0);
    if (rc == P2P_OK) rc = p2p_remap_views_pitch_maps_f64(pano.data(), pw, ph, 3 * pw, yd, 2, U.data(), V.data(), 2, 77, ow, oh, out.data(), 0);
    if (rc == P2P_OK) rc = p2p_remap_views_pitch_maps_f64(pano.data(), pw, ph, 3 * pw, yd, 2, U.data(), V.data(), 2, 77, ow, oh, out.data(), 0);
    if (rc == P2P_OK) rc = p2p_remap_maps_interp_u8(pano.data(), pw, ph, 3 * pw, 3, U.data(), V.data(), ow, oh, dst.data(), P2P_INTER_CUBIC, P2P_BORDER_WRAP, nullptr, 0);
    if (rc == P2P_OK) rc = p2p_build_yaw_row(pw, 0.3, U.data(), 0);
    g_in_library = false;
    if (rc != P2P_OK && !*p2p_last_error()) { fprintf(stderr, "a failing call left no message\n"); exit(1); }
    *rc_out = rc;
    return rc == P2P_OK;
}

// Sweep: fail the N-th allocation of a job's life for N = 1, 2, ... until a run makes fewer than N allocations.  After every
// failed run the same life must succeed with nothing armed (the library is still usable, its pools and caches consistent).
template <class F>
static void alloc_failure_sweep(const char* what, F life)
{
    int rc = 0, failed = 0, oom = 0;
    CHECK(life(&rc));   // warm: the process-wide pools exist (they are never torn down and would read as leaks of run N)
    long n = 1;
    for (;; ++n) {
        g_news = 0;
        g_fail_at = n;
        const bool ok = life(&rc);
        const bool triggered = g_news >= n;
        g_fail_at = -1;
        if (!ok) {
            ++failed;
            oom += rc == P2P_ERR_OOM;
            CHECK(rc == P2P_ERR_OOM || rc == P2P_ERR_HIP);
            CHECK(life(&rc));   // ... and the library carries on (P:279-280: an error is logged, the pool lives)
        }
        if (!triggered)
            break;
        CHECK(n < 100000);
    }
    CHECK(p2p_release_cache() == P2P_OK);
    printf("allocation-failure sweep, %s: %ld allocations per life, %d lives failed (%d with P2P_ERR_OOM), every one recovered\n", what, n - 1, failed, oom);
    fflush(stdout);
}

int main()
{
    CHECK(p2p_device_count() == 1);
    if (!getenv("P2P_SAN_NO_SWEEP")) {
        alloc_failure_sweep("resident job, per-view plan", [](int* rc) { return job_life(false, rc); });
        setenv("P2P_BAND", "1", 1);
        CHECK(p2p_reload_options() == P2P_OK);
        alloc_failure_sweep("resident job, band plan", [](int* rc) { return job_life(true, rc); });
        unsetenv("P2P_BAND");
        CHECK(p2p_reload_options() == P2P_OK);
        alloc_failure_sweep("one-shot entry points", oneshot_life);
    }
    // ---- argument errors: every one must come back as a status, nothing may be touched ----
    uint8_t px[48] = {0};
    const int32_t y0[1] = {0}, p_bad[1] = {0}, p_ok[1] = {90};
    CHECK(p2p_remap_views_u8(nullptr, 4, 4, 12, y0, 1, p_ok, 1, 90, 4, 4, px, 0, 0) == P2P_ERR_INVALID);
    CHECK(p2p_remap_views_u8(px, 4, 4, 12, y0, 1, p_bad, 1, 90, 4, 4, px, 0, 0) == P2P_ERR_INVALID);
    CHECK(p2p_remap_views_u8(px, 4, 4, 12, y0, 1, p_ok, 1, 90, 40000, 4, px, 0, 0) == P2P_ERR_INVALID);
    CHECK(p2p_remap_views_u8(px, 4, 4, 12, y0, 1, p_ok, 1, 90, 4, 4, px, 7, 0) == P2P_ERR_NO_DEVICE);
    CHECK(p2p_remap_views_u8(px, 4, 4, 12, y0, 0, p_ok, 1, 90, 4, 4, px, 0, 0) == P2P_OK);  // empty yaw list: nothing to do
    CHECK(p2p_remap_views_u8(px, 4, 4, 8, y0, 1, p_ok, 1, 90, 4, 4, px, 0, 0) == P2P_ERR_INVALID);  // row stride < 3 * pw
    p2p_ctx* ctx = nullptr;
    CHECK(p2p_ctx_create(0, nullptr) == P2P_ERR_INVALID && p2p_ctx_create(3, &ctx) == P2P_ERR_NO_DEVICE);
    {   // the cap on live contexts (a process that creates them without bound takes the GPU down): P2P_MAX_CONTEXTS
        setenv("P2P_MAX_CONTEXTS", "5", 1);
        CHECK(p2p_reload_options() == P2P_OK);   // the environment is read once per process, and again on request
        std::vector<p2p_ctx*> many;   // (the one-shot calls above may have left a slot's context alive: at most 5 fit, not exactly 5)
        int refused = 0;
        for (int i = 0; i < 8; ++i) {
            p2p_ctx* c = nullptr;
            const int rc = p2p_ctx_create(0, &c);
            CHECK((rc == P2P_OK && c != nullptr && refused == 0) || (rc == P2P_ERR_OOM && c == nullptr));
            if (c) many.push_back(c); else ++refused;
        }
        CHECK(refused >= 3 && !many.empty());
        p2p_ctx_destroy(many.back());
        many.pop_back();
        p2p_ctx* again = nullptr;
        CHECK(p2p_ctx_create(0, &again) == P2P_OK);   // a destroyed context makes room
        many.push_back(again);
        for (p2p_ctx* c : many) p2p_ctx_destroy(c);
        unsetenv("P2P_MAX_CONTEXTS");
        CHECK(p2p_reload_options() == P2P_OK);
    }
    // ---- what a context and a job cost: ONE stream and a handful of events, unless timing / async copies are asked for
    CHECK(p2p_release_cache() == P2P_OK);
    const long ev_before = p2p_stub_live(0), st_before = p2p_stub_live(1);
    CHECK(p2p_ctx_create(0, &ctx) == P2P_OK);
    p2p_job* job = nullptr;
    p2p_job_desc d = {64, 32, 2, 3, nullptr, 2, nullptr, 90, 70, 33, P2P_FLAG_DEFAULT};
    CHECK(p2p_job_create(ctx, &d, &job) == P2P_ERR_INVALID && job == nullptr);  // NULL angle lists
    const int32_t yaws[3] = {0, 90, 14}, pitches[2] = {30, 150};
    d.yaw_deg = yaws; d.pitch_deg = pitches;
    CHECK(p2p_job_create(ctx, &d, &job) == P2P_OK);
    std::vector<uint8_t> pano((size_t)64 * 32 * 3, 9), views((size_t)3 * 2 * 33 * 70 * 3);
    CHECK(p2p_job_run(job) == P2P_ERR_STATE);                        // no panorama yet
    CHECK(p2p_job_set_pano(job, 2, pano.data(), 192) == P2P_ERR_INVALID);
    CHECK(p2p_job_set_pano(job, 0, pano.data(), 192) == P2P_OK);
    CHECK(p2p_job_run(job) == P2P_ERR_STATE);                        // panorama 1 still missing
    CHECK(p2p_job_get_views(job, 0, views.data()) == P2P_ERR_STATE);
    CHECK(p2p_job_set_pano(job, 1, pano.data(), 192) == P2P_OK);
    CHECK(p2p_job_run(job) == P2P_OK);
    CHECK(p2p_job_get_views(job, 0, views.data()) == P2P_OK);
    {   // a fresh context + job + one run + one download: <= 8 events (4 of the context, 3 of the job), one stream
        p2p_job_info info;
        CHECK(p2p_job_get_info(job, &info) == P2P_OK && info.timing_events == 0 && info.copy_streams == 0);
        CHECK(info.tile_w == 64 || info.tile_w == 128);
        CHECK(p2p_stub_live(0) - ev_before <= 8 && p2p_stub_live(1) - st_before == 1);
        float ms1 = 0.0f;
        CHECK(p2p_job_kernel_ms(job, &ms1) == P2P_ERR_STATE);        // nobody asked for timing
    }
    CHECK(p2p_job_set_pano_async(job, 1, pano.data(), 192) == P2P_OK);  // the upload stream appears with its first use
    CHECK(p2p_job_wait(job) == P2P_OK && p2p_stub_live(1) - st_before == 2);
    CHECK(p2p_job_time_launches(job, -1) == P2P_ERR_INVALID && p2p_job_time_launches(job, 5000) == P2P_ERR_INVALID);
    CHECK(p2p_job_time_launches(job, 256) == P2P_OK);                // 512 events, because somebody asked
    for (int i = 0; i < 300; ++i)                                    // past the ring of 256 event pairs
        CHECK(p2p_job_run(job) == P2P_OK);
    float ms[300];
    CHECK(p2p_job_kernel_ms_last(job, ms, 256) == P2P_OK && p2p_job_kernel_ms_last(job, ms, 257) == P2P_ERR_STATE);
    CHECK(p2p_job_time_launches(job, 0) == P2P_OK && p2p_job_run(job) == P2P_OK && p2p_job_kernel_ms(job, ms) == P2P_ERR_STATE);
    CHECK(p2p_job_get_views_async(job, 1, views.data()) == P2P_OK && p2p_job_wait(job) == P2P_OK);
    {   // a sparse view set, one view at a time, and back to all views
        const uint8_t mask[6] = {1, 0, 0, 1, 1, 1};                  // [n_yaw = 3][n_pitch = 2]
        p2p_job_info info;
        CHECK(p2p_job_set_view_mask(job, mask) == P2P_OK && p2p_job_get_info(job, &info) == P2P_OK && info.n_views_wanted == 4);
        CHECK(p2p_job_run(job) == P2P_OK);
        std::vector<uint8_t> one((size_t)33 * 70 * 3);
        CHECK(p2p_job_get_view(job, 1, 2, 1, one.data()) == P2P_OK && p2p_job_get_view(job, 0, 3, 0, one.data()) == P2P_ERR_INVALID);
        CHECK(p2p_job_get_view_async(job, 0, 0, 0, one.data()) == P2P_ERR_STATE);   // 70 is not divisible by 4: packed downloads only
        CHECK(p2p_job_set_view_mask(job, nullptr) == P2P_OK && p2p_job_get_info(job, &info) == P2P_OK && info.n_views_wanted == 6);
    }
    {   // a band of rows of every view (one image's rows shared out to several GPUs), its rows back, and the whole view again
        CHECK(p2p_job_set_rows(job, 8, 32) == P2P_ERR_INVALID && p2p_job_set_rows(job, 0, 40) == P2P_ERR_INVALID);
        CHECK(p2p_job_set_rows(job, 16, 33) == P2P_OK && p2p_job_run(job) == P2P_OK);
        std::vector<uint8_t> part((size_t)17 * 70 * 3);
        CHECK(p2p_job_get_view_rows(job, 1, 2, 1, 16, 33, part.data()) == P2P_OK);
        CHECK(p2p_job_get_view_rows(job, 1, 2, 1, 16, 34, part.data()) == P2P_ERR_INVALID);
        CHECK(p2p_job_get_view_rows_async(job, 0, 0, 0, 16, 33, part.data()) == P2P_ERR_STATE);   // 70 is not divisible by 4
        CHECK(p2p_job_set_rows(job, 0, 33) == P2P_OK && p2p_job_run(job) == P2P_OK);
    }
    {   // the pitch stage's border mode (the legacy tool's resident job): bad codes refused, a change re-plans, and back
        CHECK(p2p_job_set_border(job, 7) == P2P_ERR_INVALID && p2p_job_set_border(nullptr, 0) == P2P_ERR_INVALID);
        CHECK(p2p_job_set_border(job, P2P_BORDER_REFLECT) == P2P_OK && p2p_job_run(job) == P2P_OK);
        CHECK(p2p_job_set_border(job, P2P_BORDER_CONSTANT) == P2P_OK && p2p_job_run(job) == P2P_OK);
    }
    std::vector<int32_t> coords((size_t)2 * 33 * 70 * 2);
    CHECK(p2p_job_get_coords(job, coords.data()) == P2P_OK);
    std::vector<uint32_t> tabs((size_t)3 * 64);
    CHECK(p2p_job_get_yaw_tables(job, tabs.data()) == P2P_OK);
    const int32_t yaws2[3] = {1, 2, 3};
    CHECK(p2p_job_set_yaws(job, yaws2) == P2P_OK);
    std::vector<float> rows((size_t)3 * 64, 5.0f), U((size_t)2 * 33 * 70, 1.0f), V((size_t)2 * 33 * 70, 1.0f);
    rows[7] = 64.0f;                                                 // outside [0, pw - 1]
    CHECK(p2p_job_set_maps(job, rows.data(), U.data(), V.data()) == P2P_ERR_INVALID);
    rows[7] = 63.0f;
    CHECK(p2p_job_set_maps(job, rows.data(), U.data(), V.data()) == P2P_OK && p2p_job_run(job) == P2P_OK);
    {   // a band plan (source-band tiles): its scratch, its two-stage build with the read-back in between, the launch, a
        // view mask on top, and the way back to a per-view plan when P2P_BAND goes away
        setenv("P2P_BAND", "1", 1);
        CHECK(p2p_reload_options() == P2P_OK);
        p2p_job* jb = nullptr;
        const int32_t yb[2] = {0, 90}, pb[2] = {60, 120};
        p2p_job_desc db = {64, 32, 1, 2, yb, 2, pb, 90, 70, 33, P2P_FLAG_DEFAULT};
        CHECK(p2p_job_create(ctx, &db, &jb) == P2P_OK);
        CHECK(p2p_job_set_pano(jb, 0, pano.data(), 192) == P2P_OK && p2p_job_run(jb) == P2P_OK && p2p_job_run(jb) == P2P_OK);
        p2p_job_info ib;
        CHECK(p2p_job_get_info(jb, &ib) == P2P_OK && ib.band_tiles > 0);
        const uint8_t mb[4] = {1, 0, 0, 1};
        CHECK(p2p_job_set_view_mask(jb, mb) == P2P_OK && p2p_job_run(jb) == P2P_OK);
        std::vector<uint8_t> vb((size_t)4 * 33 * 70 * 3);
        CHECK(p2p_job_get_views(jb, 0, vb.data()) == P2P_OK);
        std::vector<int32_t> cb((size_t)2 * 33 * 70 * 2);
        CHECK(p2p_job_get_coords(jb, cb.data()) == P2P_OK);
        p2p_job_destroy(jb);
        {   // ... and one whose counts never reach the host's page-locked block: read back with copies, the same plan
            p2p_stub_drop_count = 1;
            const int32_t pb2[2] = {61, 119};
            db.pitch_deg = pb2;
            p2p_job_info ib2;
            CHECK(p2p_job_create(ctx, &db, &jb) == P2P_OK && p2p_job_set_pano(jb, 0, pano.data(), 192) == P2P_OK && p2p_job_run(jb) == P2P_OK);
            CHECK(p2p_job_get_info(jb, &ib2) == P2P_OK && ib2.band_tiles == ib.band_tiles && ib2.n_gather_tiles == ib.n_gather_tiles);
            p2p_job_destroy(jb);
            p2p_stub_drop_count = 0;
        }
        unsetenv("P2P_BAND");
        CHECK(p2p_reload_options() == P2P_OK);
    }
    // a second job borrowing the first one's panoramas
    p2p_job* job2 = nullptr;
    p2p_job_desc_f64 d2 = {64, 32, 2, 1, nullptr, 1, nullptr, 60.5, 16, 16, 0};
    const double y2[1] = {12.5}, p2[1] = {77.25};
    d2.yaw_deg = y2; d2.pitch_deg = p2;
    CHECK(p2p_job_create_f64(ctx, &d2, &job2) == P2P_OK);
    CHECK(p2p_job_share_panos(job2, job) == P2P_OK && p2p_job_set_pano(job2, 0, pano.data(), 192) == P2P_ERR_STATE);
    CHECK(p2p_job_run(job2) == P2P_OK);
    std::vector<uint8_t> v2((size_t)16 * 16 * 3);
    CHECK(p2p_job_get_views(job2, 1, v2.data()) == P2P_OK);
    CHECK(p2p_job_get_view_async(job2, 1, 0, 0, v2.data()) == P2P_OK && p2p_job_wait(job2) == P2P_OK);
    // the context's table caches: a third job of job2's geometry shares its plan and yaw tables (the build times
    // reported are those of the first build), outlives it, and survives a budget of zero
    p2p_job* job3 = nullptr;
    CHECK(p2p_job_create_f64(ctx, &d2, &job3) == P2P_OK && p2p_job_share_panos(job3, job) == P2P_OK);
    float pm = -1.0f, tm = -1.0f;
    CHECK(p2p_job_plan_ms(job3, &pm, &tm) == P2P_ERR_STATE);          // not run yet
    CHECK(p2p_job_run(job3) == P2P_OK && p2p_job_plan_ms(job3, &pm, &tm) == P2P_OK && pm >= 0.0f && tm >= 0.0f);
    p2p_job_destroy(job2);
    const double y3[1] = {13.5};
    CHECK(p2p_job_set_yaws_f64(job3, y3) == P2P_OK && p2p_job_run(job3) == P2P_OK);
    std::vector<float> U3((size_t)16 * 16, 2.0f);
    CHECK(p2p_job_set_maps(job3, nullptr, U3.data(), U3.data()) == P2P_OK && p2p_job_run(job3) == P2P_OK);  // private plan
    CHECK(p2p_release_cache() == P2P_OK);                            // unused entries of THIS (explicit) context go too
    CHECK(p2p_job_run(job3) == P2P_OK);                              // ... and what a live job uses stays
    p2p_job_destroy(job3);
    {   // a context created under a cache budget of zero: every unused entry goes at the next insertion
        setenv("P2P_PLAN_CACHE_MB", "0", 1);
        CHECK(p2p_reload_options() == P2P_OK);
        p2p_ctx* c0 = nullptr;
        CHECK(p2p_ctx_create(0, &c0) == P2P_OK);
        unsetenv("P2P_PLAN_CACHE_MB");
        CHECK(p2p_reload_options() == P2P_OK);
        p2p_job *ja = nullptr, *jb = nullptr;
        CHECK(p2p_job_create_f64(c0, &d2, &ja) == P2P_OK && p2p_job_set_pano(ja, 0, pano.data(), 192) == P2P_OK &&
              p2p_job_set_pano(ja, 1, pano.data(), 192) == P2P_OK && p2p_job_run(ja) == P2P_OK);
        p2p_job_destroy(ja);
        d2.yaw_deg = y3;
        CHECK(p2p_job_create_f64(c0, &d2, &jb) == P2P_OK && p2p_job_set_pano(jb, 0, pano.data(), 192) == P2P_OK &&
              p2p_job_set_pano(jb, 1, pano.data(), 192) == P2P_OK && p2p_job_run(jb) == P2P_OK);
        d2.yaw_deg = y2;
        p2p_job_destroy(jb);
        p2p_ctx_destroy(c0);
    }
    p2p_job_destroy(job);
    p2p_job_destroy(nullptr);
    void* host = nullptr;
    CHECK(p2p_host_alloc(1 << 20, &host) == P2P_OK && p2p_host_free(host) == P2P_OK && p2p_host_free(nullptr) == P2P_OK);
    // ---- one-shot entry points from more threads than the pool has slots ----
    std::vector<std::thread> th;
    for (int i = 0; i < 12; ++i)
        th.emplace_back(one_shot_calls, i);
    for (auto& t : th)
        t.join();
    CHECK(p2p_release_cache() == P2P_OK);
    one_shot_calls(99);  // the pool's contexts survive a release
    CHECK(p2p_release_cache() == P2P_OK);
    {   // a plan pass whose count never reaches the host (a lost write on the device): the headers say which tiles gather
        p2p_stub_drop_count = 1;
        p2p_job* jd = nullptr;
        const int32_t yd[2] = {0, 77}, pd[2] = {45, 100};
        p2p_job_desc dd = {64, 32, 1, 2, yd, 2, pd, 90, 70, 33, P2P_FLAG_DEFAULT};
        std::vector<uint8_t> pano_d((size_t)64 * 32 * 3, 5), views_d((size_t)2 * 2 * 33 * 70 * 3);
        CHECK(p2p_job_create(ctx, &dd, &jd) == P2P_OK && p2p_job_set_pano(jd, 0, pano_d.data(), 192) == P2P_OK);
        CHECK(p2p_job_run(jd) == P2P_OK && p2p_job_run(jd) == P2P_OK && p2p_job_get_views(jd, 0, views_d.data()) == P2P_OK);
        p2p_job_info info;
        CHECK(p2p_job_get_info(jd, &info) == P2P_OK && info.n_gather_tiles > 0);  // (the stub's plan marks a third of the tiles)
        p2p_job_destroy(jd);
        p2p_stub_drop_count = 0;
        CHECK(p2p_job_create(ctx, &dd, &jd) == P2P_OK && p2p_job_set_pano(jd, 0, pano_d.data(), 192) == P2P_OK && p2p_job_run(jd) == P2P_OK);
        p2p_job_destroy(jd);
    }
    p2p_ctx_destroy(ctx);
    p2p_stub_device_count = 0;
    CHECK(p2p_device_count() == 0 && p2p_remap_views_u8(px, 4, 4, 12, y0, 1, p_ok, 1, 90, 4, 4, px, 0, 0) == P2P_ERR_NO_DEVICE);
    printf("host sanitizer run OK\n");
    return 0;
}
