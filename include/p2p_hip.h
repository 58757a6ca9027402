/*
 * p2p_hip.h -- C ABI of libp2p_hip.so, the MI355X (gfx950) implementation of the
 * equirectangular -> perspective view-synthesis hot path of
 * Maxiviper117/360-to-planer-images.
 *
 * The reference has NO FFI: its boundary for this path is a handful of Python
 * functions in app/panorama_to_plane-pitch.py ("P") and app/legacy/panorama_to_plane.py
 * ("L").  Each entry point below names the reference interface it replaces; the
 * Python mirror that binds them with ctypes lives in
 * 360-to-planer-images_amd/panorama_to_plane_pitch.py and the stub a reference
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; the caller owns every host buffer;
 * the library owns device memory, streams and events; every function returns
 * P2P_OK (0) or a negative p2p_status and never throws; p2p_last_error() gives a
 * thread-local message for the last failure on the calling thread.  All entry
 * points are re-entrant; the one-shot functions share a small pool of device
 * contexts, so the reference's ThreadPoolExecutor fan-out (P:252-265) can call
 * them concurrently on one shared panorama.
 *
 * There is no CPU fallback in this library: without a usable HIP device every
 * compute entry point fails with P2P_ERR_NO_DEVICE.
 */
#ifndef P2P_HIP_H
#define P2P_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum p2p_status {
    P2P_OK = 0,
    P2P_ERR_INVALID = -1,     /* bad argument (what OpenCV would CV_Assert on, or NULL/size errors) */
    P2P_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range */
    P2P_ERR_HIP = -3,         /* a HIP runtime call failed; see p2p_last_error() */
    P2P_ERR_OOM = -4,         /* device or host allocation failed */
    P2P_ERR_STATE = -5        /* call sequence error (e.g. run before a panorama was set) */
} p2p_status;

/* cv2 border codes accepted by p2p_remap_maps_u8 (OpenCV core/base.hpp numbering). */
enum { P2P_BORDER_CONSTANT = 0, P2P_BORDER_REPLICATE = 1, P2P_BORDER_REFLECT = 2,
       P2P_BORDER_WRAP = 3, P2P_BORDER_REFLECT_101 = 4 };

/* interpolation codes = cv2.INTER_NEAREST / INTER_LINEAR / INTER_CUBIC (the legacy tool's method table, L:172-176) */
enum { P2P_INTER_NEAREST = 0, P2P_INTER_LINEAR = 1, P2P_INTER_CUBIC = 2 };

/* p2p_job_desc.flags / p2p_remap_views_u8 flags */
enum {
    P2P_FLAG_DEFAULT = 0,
    /* (bits 1 and 2 are unused: every job keeps its quantised pitch-stage coordinates -- p2p_job_get_coords -- and
       evaluates its pitch maps once, in the plan pass of its first p2p_job_run, as the reference's
       pitch_mapping_cache does, P:17-18, P:62-73) */
    /* Float pixel path, opt-in and BEYOND the reference (BASELINE config 5's "fp16 pixel path"):
       one float resample per view instead of two fixed-point ones -- the pitch map's coordinate (azimuth left
       unclipped) shifted by yaw * pw / 360 with true wrap-around at the seam, no 1/32-pixel quantisation, no uint8
       intermediate; the 2x2 blend in float32, or with float16 taps and weights accumulated in float32; rounded to
       uint8 once (half-even).  Not comparable bit for bit with cv2.remap; within 1-2 levels of the exact path on
       band-limited panoramas. */
    P2P_FLAG_PIXELS_F32 = 4,
    P2P_FLAG_PIXELS_F16 = 8,
    /* With a float pixel path only: rays through the centres of the output pixels and panorama texels centred at
       i + 0.5 (the reference samples at integer coordinates, P:122-131, which shifts the picture by half a pixel). */
    P2P_FLAG_PIXEL_CENTRES = 16
};

const char* p2p_version(void);
const char* p2p_last_error(void);
/* Number of usable HIP devices (0 when there is none; never an error). */
int p2p_device_count(void);

/* ------------------------------------------------------------------------------------------
 * One-shot host-buffer API (what the Python drop-in functions bind)
 * ---------------------------------------------------------------------------------------- */

/*
 * Replaces process_yaw_and_pitchs() (P:181-221) for a whole list of yaws, i.e. the
 * per-image fan-out of process_single_image() (P:252-265): for every yaw the panorama is
 * resampled by the yaw map (P:79-108, cv2.remap P:192-199) and every pitch view is gathered
 * from that resampled panorama (P:114-175, cv2.remap P:212-218), all inside one HIP kernel.
 *   pano       : uint8 [ph][pw][3], row_stride bytes between rows (channel order untouched)
 *   out        : uint8 [n_yaw][n_pitch][oh][ow][3], contiguous
 *   yaw_deg    : any integers (P:431-437 leaves yaw unvalidated; it wraps through '%', P:98)
 *   pitch_deg  : 1..179 (check_pitch, P:362-376) -- other values are P2P_ERR_INVALID
 */
int p2p_remap_views_u8(const uint8_t* pano, int pw, int ph, int64_t row_stride,
                       const int32_t* yaw_deg, int n_yaw,
                       const int32_t* pitch_deg, int n_pitch, int fov_deg,
                       int ow, int oh, uint8_t* out, int device, int flags);

/*
 * The same with real-valued angles: process_yaw_and_pitchs() (P:181-221) and get_pitch_mapping() (P:55-73) hand
 * their yaw / pitch / FOV arguments to np.radians (P:85, P:64-68), so any finite number of degrees is legal input
 * to the reference's functions (only the CLI narrows them to integers and the pitch to 1..179, P:362-376,
 * P:406-437).  This entry point accepts what the functions accept; the integer one above keeps the CLI's checks.
 */
int p2p_remap_views_f64(const uint8_t* pano, int pw, int ph, int64_t row_stride,
                        const double* yaw_deg, int n_yaw,
                        const double* pitch_deg, int n_pitch, double fov_deg,
                        int ow, int oh, uint8_t* out, int device, int flags);

/*
 * The same two-stage synthesis with caller-supplied float32 maps instead of in-kernel ones:
 *   yaw_rows : [n_yaw][pw]      = U_yaw[0, :] of precompute_yaw_mapping (P:79-108; V_yaw[y,x] == y)
 *   U, V     : [n_pitch][oh][ow] = precompute_pitch_mapping outputs (P:114-175)
 * Given the reference's own maps the result is the reference's result bit for bit.
 */
int p2p_remap_views_maps_u8(const uint8_t* pano, int pw, int ph, int64_t row_stride,
                            const float* yaw_rows, int n_yaw,
                            const float* U, const float* V, int n_pitch,
                            int ow, int oh, uint8_t* out, int device);

/*
 * The tool's --exact route: the yaw tables are built on the device from yaw_deg (P:79-108 uses IEEE operations only:
 * bit-exact), the PITCH maps are the caller's -- what get_pitch_mapping() (P:55-73) returns on the caller's host,
 * libm's and BLAS's last bits included -- and every pixel is drawn from them by the fixed-point kernels: the
 * reference's bytes, also on noise panoramas.
 *   U, V     : [n_pitch][oh][ow] float32
 *   maps_key : the caller's name for exactly these maps -- the reference's pitch_mapping_cache keys them by
 *              (output_width, output_height, pitch_angle, pano_width, pano_height, fov_deg) (P:62) and keeps them for
 *              the life of the process; a one-shot slot that already holds maps of this name keeps their device copy
 *              and the plan made from them (a later call with the same key uploads the panorama only).  0: no name,
 *              the maps are uploaded and planned on every call.  Two different sets of maps must never share a key.
 */
int p2p_remap_views_pitch_maps_f64(const uint8_t* pano, int pw, int ph, int64_t row_stride,
                                   const double* yaw_deg, int n_yaw,
                                   const float* U, const float* V, int n_pitch, uint64_t maps_key,
                                   int ow, int oh, uint8_t* out, int device);

/*
 * Replaces panorama_to_plane(pano_array, U, V) (L:182-194) == cv2.remap(src, U, V,
 * INTER_LINEAR, borderMode) (L:179 uses BORDER_REFLECT; P:192-199/212-218 use BORDER_CONSTANT
 * with borderValue 0).  src: uint8 [sh][sw][cn], cn in {1,3,4}; U,V: float32 [oh][ow];
 * out: uint8 [oh][ow][cn].  border_value: cn bytes or NULL (zeros).
 */
int p2p_remap_maps_u8(const uint8_t* src, int sw, int sh, int64_t row_stride, int cn,
                      const float* U, const float* V, int ow, int oh, uint8_t* out,
                      int border_mode, const uint8_t* border_value, int device);

/*
 * The legacy tool's per-image loop, L:259-281: one image, one panorama_to_plane(pano, U_yaw, V_yaw) per yaw angle
 * with maps precomputed up front (L:359-370).  All n_maps remaps of a 3-channel image in one call: the image is
 * uploaded once and every map is drawn by the same launch.  U, V: float32 [n_maps][oh][ow]; out: uint8
 * [n_maps][oh][ow][3]; INTER_LINEAR; border_mode as p2p_remap_maps_u8 with a zero constant border.
 */
int p2p_remap_maps_batch_u8(const uint8_t* src, int sw, int sh, int64_t row_stride,
                            const float* U, const float* V, int n_maps, int ow, int oh, uint8_t* out,
                            int border_mode, int device);

/* interpolate_color(U, V, img, method) of the legacy tool, L:159-180: the same call with the method chosen,
   cv2.remap(img, U, V, interpolation, borderMode).  INTER_NEAREST rounds the coordinates half-even and copies;
   INTER_CUBIC uses OpenCV's 4x4 fixed-point kernel (A = -0.75, 1/32-pixel phases, 15-bit weights). */
int p2p_remap_maps_interp_u8(const uint8_t* src, int sw, int sh, int64_t row_stride, int cn,
                             const float* U, const float* V, int ow, int oh, uint8_t* out,
                             int interpolation, int border_mode, const uint8_t* border_value, int device);

/*
 * Replaces precompute_pitch_mapping(W, H, FOV_rad, pitch_radian, pano_width, pano_height)
 * (P:114-175; get_pitch_mapping P:55-73 passes np.radians() of its degree arguments): float32
 * U, V [oh][ow] computed on the device with the arithmetic the fused kernel uses.
 */
int p2p_build_pitch_map(int ow, int oh, double fov_rad, double pitch_rad, int pw, int ph,
                        float* U, float* V, int device);

/* The legacy tool's combined map (app/legacy/panorama_to_plane.py precompute_mapping, L:47-157): the
   normalised pinhole ray of every output pixel is rotated by the float32 3x3 matrix R9 (row-major; the
   caller builds it as get_rotation_matrix does, L:21-45: R_pitch @ R_yaw) and mapped to panorama
   coordinates, clipped to [0, pw-1] x [0, ph-1].  U, V: float32 [oh][ow] host buffers. */
int p2p_build_rot_map(int ow, int oh, double fov_rad, const float* R9, int pw, int ph,
                      float* U, float* V, int device);

/*
 * Replaces get_yaw_mapping()/precompute_yaw_mapping() (P:42-52, P:79-108): the float32 row
 * U_yaw[0, :] of length pw (every row of the reference's U equals it; V_yaw[y, x] == y).
 * yaw_rad = np.radians(yaw_angle) (P:85).
 */
int p2p_build_yaw_row(int pw, double yaw_rad, float* U_row, int device);

/* ------------------------------------------------------------------------------------------
 * Resident (device-buffer) API: the batch driver and bench.py keep panoramas and views in HBM
 * ---------------------------------------------------------------------------------------- */

typedef struct p2p_ctx p2p_ctx;   /* one device + one HIP stream (+ two copy streams once it copies asynchronously) + four events */
typedef struct p2p_job p2p_job;   /* n_panos panoramas of one size x (yaw x pitch) view set */

typedef struct p2p_job_desc {
    int32_t pw, ph;               /* panorama size (both < 32767: cv::remap's SHRT_MAX assert) */
    int32_t n_panos;              /* panoramas resident at once */
    int32_t n_yaw;
    const int32_t* yaw_deg;
    int32_t n_pitch;
    const int32_t* pitch_deg;
    int32_t fov_deg, ow, oh;
    int32_t flags;
} p2p_job_desc;

/* p2p_job_desc with real-valued angles (see p2p_remap_views_f64) */
typedef struct p2p_job_desc_f64 {
    int32_t pw, ph;
    int32_t n_panos;
    int32_t n_yaw;
    const double* yaw_deg;
    int32_t n_pitch;
    const double* pitch_deg;
    double fov_deg;
    int32_t ow, oh;
    int32_t flags;
} p2p_job_desc_f64;

/* A context owns one HIP stream and four events; its two copy streams are created by the first
   p2p_job_set_pano_async / p2p_job_get_views_async on it.  It also keeps, by the reference's cache keys, the device
   tables of every geometry its jobs have used (yaw_mapping_cache / pitch_mapping_cache, P:17-18), up to
   P2P_PLAN_CACHE_MB (default 4096; read when the context is created).  At most P2P_MAX_CONTEXTS (default 64: what
   has been run on a GPU) contexts are alive per process; beyond that p2p_ctx_create returns P2P_ERR_OOM.
   Destroying the process's last context returns the library's idle device memory to the driver. */
int p2p_ctx_create(int device, p2p_ctx** out);
void p2p_ctx_destroy(p2p_ctx* ctx);
int p2p_ctx_synchronize(p2p_ctx* ctx);
/* Two HIP event marks on the context's stream (which = 0 / 1) and the device time between them: brackets a
   whole region of launches without synchronising inside it. */
int p2p_ctx_mark(p2p_ctx* ctx, int which);
int p2p_ctx_marked_ms(p2p_ctx* ctx, float* ms);

int p2p_job_create(p2p_ctx* ctx, const p2p_job_desc* desc, p2p_job** out);
int p2p_job_create_f64(p2p_ctx* ctx, const p2p_job_desc_f64* desc, p2p_job** out);
void p2p_job_destroy(p2p_job* job);
/* H2D copy of panorama `index` (uint8 [ph][pw][3]) on the context's kernel stream, in order with its launches;
   returns once the host buffer may be reused or freed. */
int p2p_job_set_pano(p2p_job* job, int index, const uint8_t* pano, int64_t row_stride);
/* The same without waiting (cv2.imread of the NEXT image, P:244, overlaps the resampling of this one): the copy runs
   on the context's upload stream behind the job's last kernel; `pano` must stay valid and unchanged until
   p2p_job_wait / p2p_ctx_synchronize.  The job's next p2p_job_run waits for it on the device, not on the host.
   A driver that keeps TWO jobs per device and alternates them gets upload k+1 and download k-1 under kernel k. */
int p2p_job_set_pano_async(p2p_job* job, int index, const uint8_t* pano, int64_t row_stride);
/* Let `job` read the device panoramas of `owner` (same context, panorama size and count) instead of holding a copy
   of its own: the reference shares ONE pano_image among its per-yaw tasks (P:252-265); a device that draws several
   (pitch, yaw subset) groups of one image -- the view-sharded multi-GPU path -- uploads it once.  `owner` must
   outlive `job`; panoramas are set on the owner. */
int p2p_job_share_panos(p2p_job* job, p2p_job* owner);
/* Replace the job's yaw list (same count as at creation) and rebuild its column tables -- the key change
   the reference's yaw_mapping_cache sees between two process_yaw_and_pitchs calls on one image size
   (P:42-52: key (pano_width, pano_height, yaw_angle)); panoramas and pitch constants stay resident. */
int p2p_job_set_yaws(p2p_job* job, const int32_t* yaw_deg);
int p2p_job_set_yaws_f64(p2p_job* job, const double* yaw_deg);
/* Optional: use caller float maps instead of in-kernel ones (see p2p_remap_views_maps_u8).  yaw_rows may be
   NULL to keep the yaw tables built from yaw_deg. */
int p2p_job_set_maps(p2p_job* job, const float* yaw_rows, const float* U, const float* V);
/* The border mode of the job's pitch stage: P2P_BORDER_CONSTANT (0, the default: the current tool's cv2.remap calls,
   P:192-199, P:212-218) or one of the other cv2 codes -- the legacy tool's panorama_to_plane is
   cv2.remap(img, U, V, INTER_LINEAR, BORDER_REFLECT) (L:179).  With p2p_job_set_maps (its maps as the job's "pitch views",
   one per yaw of L:259-265, a single yaw of 0 degrees in front) the legacy per-image loop becomes a resident job: the
   maps go up once, every image costs one upload, one launch and one download.  The job's next run plans for the mode. */
int p2p_job_set_border(p2p_job* job, int border_mode);
/* Sparse view sets.  mask: uint8 [n_yaw][n_pitch], non-zero = the job draws that (yaw, pitch) view of every panorama;
   NULL = all of them again.  Views that are not wanted are neither computed nor written (their part of the output
   block keeps whatever it held).  What it is for: the view-sharded multi-GPU path -- one task per yaw on ONE shared
   pano_image in the reference (P:252-265); here the 36 views of an image dealt round-robin to 8 GPUs give a rank 4 or
   5 (yaw, pitch) combinations that are no full yaw x pitch grid, and with a mask they are ONE job and one launch,
   whose pitch views share the source rows they read. */
int p2p_job_set_view_mask(p2p_job* job, const uint8_t* mask);
/* Enqueue the view-synthesis kernel for all panoramas x yaws x pitches (asynchronous). */
int p2p_job_run(p2p_job* job);
/* Copy all views of panorama `index` to host, uint8 [n_yaw][n_pitch][oh][ow][3], on the kernel stream behind the
   job's last run; returns when `out` is complete. */
int p2p_job_get_views(p2p_job* job, int index, uint8_t* out);
/* The same without waiting (cv2.imwrite of image k, P:277, overlaps the resampling of image k+1): the copy runs on
   the context's download stream behind the job's last run; `out` is complete after p2p_job_wait /
   p2p_ctx_synchronize.  The job's next run waits for the copy on the device. */
int p2p_job_get_views_async(p2p_job* job, int index, uint8_t* out);
/* ONE view of panorama `index` -- yaw yaw_i, pitch pitch_i of the job's lists -- to host, uint8 [oh][ow][3]; returns
   when `out` is complete.  (Config 4 holds 18 GB of views per panorama; a caller that wants five of them.) */
int p2p_job_get_view(p2p_job* job, int index, int yaw_i, int pitch_i, uint8_t* out);
/* The same without waiting, on the context's download stream behind the job's last run (view widths divisible by 4);
   `out` is complete after p2p_job_wait / p2p_ctx_synchronize. */
int p2p_job_get_view_async(p2p_job* job, int index, int yaw_i, int pitch_i, uint8_t* out);

/* One image's ROWS shared out to several GPUs (no reference counterpart: P:252-265 fans one image's yaws out to threads;
 * with fewer images than GPUs this build deals an image's views -- or, with these calls, a band of rows of EVERY view --
 * to the devices).  p2p_job_set_rows: the job draws only output rows [row0, row1) of each of its views; row0 and row1
 * are multiples of 16 (tile rows), row1 may also be the view height; (0, oh) restores the whole view.  The job's next
 * run plans for that range.  p2p_job_get_view_rows[_async]: rows [row0, row1) of one view, packed 3 * ow bytes per row,
 * into `out`; the asynchronous form needs a view width divisible by 4 and is waited for with p2p_job_wait. */
int p2p_job_set_rows(p2p_job* job, int row0, int row1);
int p2p_job_get_view_rows(p2p_job* job, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out);
int p2p_job_get_view_rows_async(p2p_job* job, int index, int yaw_i, int pitch_i, int row0, int row1, uint8_t* out);
/* Wait for everything the job has in flight: uploads, its last run, downloads. */
int p2p_job_wait(p2p_job* job);
/* Launch timing, off by default (a job that nobody times creates no timing event and records none).  n >= 1: every
   p2p_job_run from now on brackets its kernels with its own HIP event pair and the job keeps the pairs of the last n
   launches (n <= 4096; 2 n events, created here) for p2p_job_kernel_ms / p2p_job_kernel_ms_last; n = 0: off again.
   Either call restarts the history. */
int p2p_job_time_launches(p2p_job* job, int n);
/* Device time it took to build the job's plan (the pitch maps' tables, once per geometry: the reference's
   pitch_mapping_cache, P:17-18, P:55-73) and its yaw tables (yaw_mapping_cache, P:42-52), in ms.  The context keeps
   both by geometry, so these are the times of whichever job built them first.  After the first p2p_job_run.  The plan
   pass is timed only when the job that built the plan had launch timing on (p2p_job_time_launches before its first
   run: two events around the pass, 10 us of a cold image's device time); otherwise *plan_ms is 0. */
int p2p_job_plan_ms(p2p_job* job, float* plan_ms, float* tables_ms);
/* Device time of the last p2p_job_run's view kernel(s), from HIP events on the job's stream. */
int p2p_job_kernel_ms(p2p_job* job, float* ms);
/* Device times of the last n p2p_job_run launches (n <= what p2p_job_time_launches asked for), oldest first;
   synchronises once. */
int p2p_job_kernel_ms_last(p2p_job* job, float* ms, int n);
/* Device address / byte size of the output block [n_panos][n_yaw][n_pitch][oh] rows.  A device row holds the view
   row's ow pixels (3 bytes each) padded to whole 4-pixel groups: 12 * ceil(ow / 4) bytes, = 3 * ow when ow is
   divisible by 4 -- every row then starts dword-aligned and the kernels write 12 bytes per lane for any width;
   p2p_job_get_views copies rows out into the caller's contiguous [..][oh][ow][3] array. */
void* p2p_job_device_out(p2p_job* job, int64_t* bytes);
/* The pitch-stage coordinates in 1/32 px the job's plan holds (after the first p2p_job_run),
   int32 [n_pitch][oh][ow][2] = (sx, sy); INT32_MIN marks a NaN coordinate (black pixel).  Jobs of the float
   pixel path hold float coordinates there instead (U - centre, V - centre as float32 bits). */
int p2p_job_get_coords(p2p_job* job, int32_t* sxsy);
/* The packed per-column yaw tables in use, uint32 [n_yaw][pw] = 3*ix | fx << 20. */
int p2p_job_get_yaw_tables(p2p_job* job, uint32_t* packed);
/* How the job is drawn: tile shape, pairs per workgroup, work-list order, what the plan found.  For tests and tools
   ("the test names the shape it ran"); nothing here is needed to use the library. */
typedef struct p2p_job_info {
    int32_t tile_w, tile_h;        /* output tile of one workgroup: 64 x 16 or 128 x 16 */
    int32_t pairs_per_block;       /* (panorama, yaw) pairs one workgroup loops over */
    int32_t pair_chunks;           /* chunks of pairs = ceil(n_panos * n_yaw / pairs_per_block) */
    int32_t list_order;            /* 1: the main kernel's tiles are drawn in source-band order from per-XCD lists */
    int32_t main_group;            /* list entries an XCD draws for one chunk before it turns to the next chunk */
    int32_t prefetch_lead;         /* > 0: table-prefetch workgroups, this many groups ahead */
    int32_t n_odd_yaws;            /* yaws that are not a plain shift with one weight (rest / table kernels) */
    int64_t n_tiles;               /* tiles of all pitch views */
    int64_t n_gather_tiles;        /* of those, drawn by the gather kernel (-1 before the first p2p_job_run) */
    int32_t timing_events;         /* HIP events of the launch-timing ring (0 unless p2p_job_time_launches asked) */
    int32_t copy_streams;          /* copy streams the job's context has created so far (0..2) */
    int32_t n_views_wanted;        /* views per panorama the job draws (n_yaw * n_pitch unless p2p_job_set_view_mask) */
    int32_t chunks_per_workgroup;  /* chunks of pairs one main-kernel workgroup draws in turn (1 unless the plan tables
                                      of a launch are too big to stay cached: config 4) */
    int32_t band_tiles;            /* > 0: the job is drawn from source-band tiles (this many) instead of the main
                                      kernel's per-view tiles; -1: it will be, the plan is not built yet; 0: no */
    int32_t lds_items_cap;         /* items (4 source pixels each) of one LDS buffer of the job's tile shape: 704 (64-wide
                                      tiles), 1408 (128-wide), 960 (the 64-wide shape of band jobs) */
} p2p_job_info;
int p2p_job_get_info(p2p_job* job, p2p_job_info* out);

/* ------------------------------------------------------------------------------------------
 * Host memory and the one-shot cache
 *
 * The reference keeps its maps for the life of the process (yaw_mapping_cache / pitch_mapping_cache,
 * P:17-18) and lets NumPy / cv2.imread allocate pageable arrays (P:244, the slices cv2.remap returns at
 * P:212-218).  Here the one-shot entry points run on a small pool of device contexts (P2P_ONESHOT_SLOTS per
 * device, default 4) shared by all calling threads -- the reference's fan-out is int(0.9 * cores) threads
 * (P:304-306), and a stream plus device buffers per THREAD would not scale.  Each slot keeps the device buffers,
 * tables and plan of the last call it served: a later call with the same geometry (panorama size, yaw count,
 * pitch list, FOV, output size) is handed that slot, re-uploads only the panorama and rebuilds the yaw tables if
 * the yaw values changed.  Callers beyond the pool size wait their turn.  P2P_ONESHOT_CACHE=0 turns the keeping
 * off; P2P_ONESHOT_CACHE_MAX_MB (default 4096) bounds what one slot keeps, so a device holds at most
 * slots x that.  Nothing is torn down at process exit (no HIP call after the runtime's own shutdown).
 *
 * Device memory: every buffer of the library comes from a per-device pool that keeps up to P2P_POOL_MB (default
 * 8192) of idle blocks instead of returning them to the driver.  p2p_release_cache gives back what can be given
 * back without disturbing work in flight: the cached jobs of idle one-shot slots, the cached tables and plans of
 * EVERY live context that no job uses any more, and all idle blocks of the pool; the calling thread's current
 * device is left as it was.  An allocation that fails for lack of device memory does the same once and retries.
 *
 * Environment: the P2P_* variables (INTEGRATION.md lists them) are read once per process, at the first call that
 * needs one, and copied into a job when it is created -- no entry point reads the environment on a launch path.
 * p2p_reload_options reads them again; call it only while no other thread is inside the library (tests and tools
 * that flip a knob between two jobs do).
 *
 * p2p_host_alloc returns page-locked host memory: panoramas decoded into it and views copied back into it
 * move by DMA at PCIe rate instead of through the runtime's pageable staging.
 * Any host pointer is accepted everywhere; page-locked ones are merely faster.
 * ------------------------------------------------------------------------------------------ */
int p2p_host_alloc(size_t bytes, void** out);
int p2p_host_free(void* ptr);
int p2p_release_cache(void);
int p2p_reload_options(void);
/* Free and total device memory as the driver sees them (hipMemGetInfo): what p2p_release_cache gives back shows here. */
int p2p_device_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* P2P_HIP_H */
