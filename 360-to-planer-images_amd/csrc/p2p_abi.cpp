// p2p_abi.cpp -- every extern "C" entry point of include/p2p_hip.h, and nothing else: each hands its arguments to its
// implementation (p2p_xyz -> p2p_host::xyz, p2p_host.h) behind ONE exception barrier.
//
// include/p2p_hip.h promises that no entry point throws ("no exceptions cross the ABI"): the callers are
// ctypes / cgo-style bindings, through which a C++ exception is a dead process -- the opposite of the reference, which logs
// a failed yaw and carries on (P:271-280).  The implementations use std::vector / std::map / std::shared_ptr /
// std::function freely, so a host allocation failure can surface anywhere inside them; here it becomes P2P_ERR_OOM (any
// other exception P2P_ERR_HIP) with p2p_last_error() set, and the library's state stays usable: every resource inside is
// owned by a guard (tests/sanitize/host_san_main.cpp sweeps an allocation failure over every allocation of a job's life).
#include "p2p_host.h"

#include <exception>

#define P2P_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

template <class F>
int guarded(const char* fn, F&& f) noexcept
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return p2p_host::fail(P2P_ERR_OOM, "%s: host allocation failed", fn);
    } catch (const std::exception& e) {
        return p2p_host::fail(P2P_ERR_HIP, "%s: %s", fn, e.what());
    } catch (...) {
        return p2p_host::fail(P2P_ERR_HIP, "%s: unknown exception", fn);
    }
}

template <class F>
void guarded_void(const char* fn, F&& f) noexcept
{
    (void)guarded(fn, [&] { f(); return 0; });
}

template <class F>
void* guarded_ptr(const char* fn, F&& f) noexcept
{
    void* p = nullptr;
    (void)guarded(fn, [&] { p = f(); return 0; });
    return p;
}

}  // namespace

#define P2P_ABI_INT(name, params, args) \
    P2P_EXPORT int p2p_##name params { return guarded("p2p_" #name, [&] { return p2p_host::name args; }); }
#define P2P_ABI_VOID(name, params, args) \
    P2P_EXPORT void p2p_##name params { guarded_void("p2p_" #name, [&] { p2p_host::name args; }); }
#define P2P_ABI_PTR(name, params, args) \
    P2P_EXPORT void* p2p_##name params { return guarded_ptr("p2p_" #name, [&] { return p2p_host::name args; }); }

// (these three cannot fail and allocate nothing)
P2P_EXPORT const char* p2p_version(void) { return p2p_host::version(); }
P2P_EXPORT const char* p2p_last_error(void) { return p2p_host::last_error(); }
P2P_EXPORT int p2p_device_count(void) { return p2p_host::device_count(); }

P2P_ABI_INT(remap_views_u8, (const uint8_t* pano, int pw, int ph, int64_t row_stride, const int32_t* yaw_deg, int n_yaw,
                             const int32_t* pitch_deg, int n_pitch, int fov_deg, int ow, int oh, uint8_t* out, int device,
                             int flags),
            (pano, pw, ph, row_stride, yaw_deg, n_yaw, pitch_deg, n_pitch, fov_deg, ow, oh, out, device, flags))
P2P_ABI_INT(remap_views_f64, (const uint8_t* pano, int pw, int ph, int64_t row_stride, const double* yaw_deg, int n_yaw,
                              const double* pitch_deg, int n_pitch, double fov_deg, int ow, int oh, uint8_t* out,
                              int device, int flags),
            (pano, pw, ph, row_stride, yaw_deg, n_yaw, pitch_deg, n_pitch, fov_deg, ow, oh, out, device, flags))
P2P_ABI_INT(remap_views_maps_u8, (const uint8_t* pano, int pw, int ph, int64_t row_stride, const float* yaw_rows,
                                  int n_yaw, const float* U, const float* V, int n_pitch, int ow, int oh, uint8_t* out,
                                  int device),
            (pano, pw, ph, row_stride, yaw_rows, n_yaw, U, V, n_pitch, ow, oh, out, device))
P2P_ABI_INT(remap_views_pitch_maps_f64, (const uint8_t* pano, int pw, int ph, int64_t row_stride, const double* yaw_deg,
                                         int n_yaw, const float* U, const float* V, int n_pitch, uint64_t maps_key,
                                         int ow, int oh, uint8_t* out, int device),
            (pano, pw, ph, row_stride, yaw_deg, n_yaw, U, V, n_pitch, maps_key, ow, oh, out, device))
P2P_ABI_INT(remap_maps_u8, (const uint8_t* src, int sw, int sh, int64_t row_stride, int cn, const float* U,
                            const float* V, int ow, int oh, uint8_t* out, int border_mode, const uint8_t* border_value,
                            int device),
            (src, sw, sh, row_stride, cn, U, V, ow, oh, out, border_mode, border_value, device))
P2P_ABI_INT(remap_maps_batch_u8, (const uint8_t* src, int sw, int sh, int64_t row_stride, const float* U, const float* V,
                                  int n_maps, int ow, int oh, uint8_t* out, int border_mode, int device),
            (src, sw, sh, row_stride, U, V, n_maps, ow, oh, out, border_mode, device))
P2P_ABI_INT(remap_maps_interp_u8, (const uint8_t* src, int sw, int sh, int64_t row_stride, int cn, const float* U,
                                   const float* V, int ow, int oh, uint8_t* out, int interpolation, int border_mode,
                                   const uint8_t* border_value, int device),
            (src, sw, sh, row_stride, cn, U, V, ow, oh, out, interpolation, border_mode, border_value, device))
P2P_ABI_INT(build_pitch_map, (int ow, int oh, double fov_rad, double pitch_rad, int pw, int ph, float* U, float* V,
                              int device),
            (ow, oh, fov_rad, pitch_rad, pw, ph, U, V, device))
P2P_ABI_INT(build_rot_map, (int ow, int oh, double fov_rad, const float* R9, int pw, int ph, float* U, float* V,
                            int device),
            (ow, oh, fov_rad, R9, pw, ph, U, V, device))
P2P_ABI_INT(build_yaw_row, (int pw, double yaw_rad, float* U_row, int device), (pw, yaw_rad, U_row, device))
P2P_ABI_INT(ctx_create, (int device, p2p_ctx** out), (device, out))
P2P_ABI_VOID(ctx_destroy, (p2p_ctx* ctx), (ctx))
P2P_ABI_INT(ctx_synchronize, (p2p_ctx* ctx), (ctx))
P2P_ABI_INT(ctx_mark, (p2p_ctx* ctx, int which), (ctx, which))
P2P_ABI_INT(ctx_marked_ms, (p2p_ctx* ctx, float* ms), (ctx, ms))
P2P_ABI_INT(job_create, (p2p_ctx* ctx, const p2p_job_desc* desc, p2p_job** out), (ctx, desc, out))
P2P_ABI_INT(job_create_f64, (p2p_ctx* ctx, const p2p_job_desc_f64* desc, p2p_job** out), (ctx, desc, out))
P2P_ABI_VOID(job_destroy, (p2p_job* job), (job))
P2P_ABI_INT(job_set_pano, (p2p_job* job, int index, const uint8_t* pano, int64_t row_stride),
            (job, index, pano, row_stride))
P2P_ABI_INT(job_set_pano_async, (p2p_job* job, int index, const uint8_t* pano, int64_t row_stride),
            (job, index, pano, row_stride))
P2P_ABI_INT(job_share_panos, (p2p_job* job, p2p_job* owner), (job, owner))
P2P_ABI_INT(job_set_yaws, (p2p_job* job, const int32_t* yaw_deg), (job, yaw_deg))
P2P_ABI_INT(job_set_yaws_f64, (p2p_job* job, const double* yaw_deg), (job, yaw_deg))
P2P_ABI_INT(job_set_maps, (p2p_job* job, const float* yaw_rows, const float* U, const float* V), (job, yaw_rows, U, V))
P2P_ABI_INT(job_set_border, (p2p_job* job, int border_mode), (job, border_mode))
P2P_ABI_INT(job_set_view_mask, (p2p_job* job, const uint8_t* mask), (job, mask))
P2P_ABI_INT(job_run, (p2p_job* job), (job))
P2P_ABI_INT(job_get_views, (p2p_job* job, int index, uint8_t* out), (job, index, out))
P2P_ABI_INT(job_get_views_async, (p2p_job* job, int index, uint8_t* out), (job, index, out))
P2P_ABI_INT(job_get_view, (p2p_job* job, int index, int yaw_i, int pitch_i, uint8_t* out),
            (job, index, yaw_i, pitch_i, out))
P2P_ABI_INT(job_get_view_async, (p2p_job* job, int index, int yaw_i, int pitch_i, uint8_t* out),
            (job, index, yaw_i, pitch_i, out))
P2P_ABI_INT(job_set_rows, (p2p_job* job, int row0, int row1), (job, row0, row1))
P2P_ABI_INT(job_get_view_rows, (p2p_job* job, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out),
            (job, index, yaw_i, pitch_i, row0, row1, out))
P2P_ABI_INT(job_get_view_rows_async, (p2p_job* job, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out),
            (job, index, yaw_i, pitch_i, row0, row1, out))
P2P_ABI_INT(job_wait, (p2p_job* job), (job))
P2P_ABI_INT(job_time_launches, (p2p_job* job, int n), (job, n))
P2P_ABI_INT(job_plan_ms, (p2p_job* job, float* plan_ms, float* tables_ms), (job, plan_ms, tables_ms))
P2P_ABI_INT(job_kernel_ms, (p2p_job* job, float* ms), (job, ms))
P2P_ABI_INT(job_kernel_ms_last, (p2p_job* job, float* ms, int n), (job, ms, n))
P2P_ABI_PTR(job_device_out, (p2p_job* job, int64_t* bytes), (job, bytes))
P2P_ABI_INT(job_get_coords, (p2p_job* job, int32_t* sxsy), (job, sxsy))
P2P_ABI_INT(job_get_yaw_tables, (p2p_job* job, uint32_t* packed), (job, packed))
P2P_ABI_INT(job_get_info, (p2p_job* job, p2p_job_info* out), (job, out))
P2P_ABI_INT(host_alloc, (size_t bytes, void** out), (bytes, out))
P2P_ABI_INT(host_free, (void* ptr), (ptr))
P2P_ABI_INT(release_cache, (void), ())
P2P_ABI_INT(reload_options, (void), ())
P2P_ABI_INT(device_mem_info, (int device, int64_t* free_bytes, int64_t* total_bytes), (device, free_bytes, total_bytes))
