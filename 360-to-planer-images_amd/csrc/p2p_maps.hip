// p2p_maps.hip -- coordinate-map kernels: yaw tables and descriptors, pitch map, the legacy tool's combined-rotation map
//   yaw map            P:79-108   -> yaw_table_kernel (bit-exact dtype flow: f32, then f64), yaw_desc_kernel
//   pitch map          P:114-175  -> pitch_map_kernel (pitch_map_eval of p2p_inline.h, f32)
//   precompute_mapping L:47-157   -> rot_map_kernel (combined yaw + pitch rotation, f32)
// Reference behaviour (cited, never copied):
//   P = /root/reference/app/panorama_to_plane-pitch.py, L = /root/reference/app/legacy/panorama_to_plane.py
// The fixed-point arithmetic is OpenCV 4.10's (imgwarp.cpp remapBilinear, INTER_BITS = 5,
// INTER_REMAP_COEF_BITS = 15); see DESIGN.md "Arithmetic contract".
// Compiled with -ffp-contract=off: every float operation below rounds where NumPy rounds.
#include "p2p_inline.h"

namespace p2p {

// ---------------------------------------------------------------------------------------------
// yaw tables: P:79-108 per column, then the cv::remap quantisation of that coordinate
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float yaw_row_eval(int col, int pw, double yaw_rad)
{
    const float TWO_PI_F = 6.283185307179586f;
    const double TWO_PI_D = 6.283185307179586;
    float u = (float)col;
    float phi = __fdiv_rn(TWO_PI_F * u, (float)pw);  // P:95 (float32)
    double pr = (double)phi + yaw_rad;                // P:98: float32 + np.float64 -> float64
    // NumPy's floored '%': fmod, then the sign fix.  fmod is exact by definition; for 0 <= pr < 4 pi -- every column of a
    // yaw in [0, 360) degrees -- so are pr itself and pr - 2 pi (y / 2 <= x <= 2 y: the difference of two doubles that close
    // is representable), and the general routine, a long loop in double precision, is only entered by lanes outside that range
    double m;
    if (pr >= 0.0 && pr < 2.0 * TWO_PI_D)
        m = pr >= TWO_PI_D ? pr - TWO_PI_D : pr;
    else
        m = fmod(pr, TWO_PI_D);
    if (m != 0.0) {
        if (m < 0.0)
            m += TWO_PI_D;
    } else {
        m = 0.0;
    }
    double Ud = __ddiv_rn(m * (double)pw, TWO_PI_D);  // P:101
    double hi = (double)(pw - 1);
    Ud = Ud < 0.0 ? 0.0 : (Ud > hi ? hi : Ud);        // P:105
    return (float)Ud;                                 // .astype(np.float32)
}

__device__ __forceinline__ uint32_t pack_yaw_entry(float U)
{
    int sx = cv_round_f32(U * 32.0f);
    int ix = sat_short(sx >> 5);
    int fx = sx & 31;
    if (ix < 0) { ix = 0; fx = 0; }  // unreachable for clipped maps; keeps the gather in bounds
    return (uint32_t)(3 * ix) | ((uint32_t)fx << 20);
}

__global__ void yaw_table_kernel(uint32_t* __restrict__ packed, float* __restrict__ rows,
                                 int pw, const double* __restrict__ yaw_rad)
{
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    int yi = blockIdx.y;
    if (col >= pw)
        return;
    float U = yaw_row_eval(col, pw, yaw_rad[yi]);
    if (rows)
        rows[(size_t)yi * pw + col] = U;
    if (packed)
        packed[(size_t)yi * pw + col] = pack_yaw_entry(U);
}

// caller-supplied float rows (p2p_job_set_maps) -> packed tables
__global__ void yaw_pack_kernel(uint32_t* __restrict__ packed, const float* __restrict__ rows, size_t n)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n)
        packed[k] = pack_yaw_entry(rows[k]);
}

// ---------------------------------------------------------------------------------------------
// Yaw descriptor: the yaw table of P:79-108 is a circular shift.  For rot column c the source
// column is (c + s) mod pw and the two-tap weight is F[c] in 0..32 (F == 32 encodes "next pixel,
// fraction 0"; (32-F)*a + F*b + 16 >> 5 then returns b exactly, which is what the table's entry
// (i+1, 0) gives).  F is one value for the whole yaw except (a) the single column that P:105
// clips to pw-1 and (b) yaws whose shift fraction sits within float noise of a 1/32-px rounding
// tie, where F flickers between two neighbours column by column.  mode 0: uniform F (+ optional
// clamp column); mode 1: per-column F (f4tab); mode 2: not a shift at all (only possible with
// caller-supplied rows) -> the kernel's direct path uses the packed table.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int yaw_delta(uint32_t te, int c, int s, int pw)
{
    int off = (int)(te & 0xFFFFFu), f = (int)(te >> 20);
    int col = c + s;
    if (col >= pw)
        col -= pw;
    if (off == 3 * col)
        return f;
    if (off == 3 * (col + 1) && f == 0 && col + 1 < pw)
        return 32;
    return -1;
}

// (1024 threads per yaw: the two passes over the table are chains of dependent loads, 8 rounds each at 8192 columns)
// EVAL: the packed table is not read but MADE here, in the first pass (yaw_table_kernel's arithmetic, entry for entry) --
// the two kernels were a launch and a dependency apart (3 + 6 us of the 30 a job's yaw tables cost a cold image).
template <bool EVAL>
__global__ __launch_bounds__(1024) void yaw_desc_kernel(YawDesc* __restrict__ desc, uint32_t* __restrict__ f4tab,
                                                       uint32_t* packed, int pw, const double* __restrict__ yaw_rad)
{
    __shared__ int bad[2];
    __shared__ int dmin, dmax;
    const int yi = blockIdx.x, t = threadIdx.x;
    uint32_t* T = packed + (size_t)yi * pw;
    uint32_t* F4 = f4tab + (size_t)yi * pw;
    const double yr = EVAL ? yaw_rad[yi] : 0.0;
    const int i0 = (int)((EVAL ? pack_yaw_entry(yaw_row_eval(0, pw, yr)) : T[0]) & 0xFFFFFu) / 3;
    if (t < 2)
        bad[t] = 0;
    if (t == 0) {
        dmin = 64;
        dmax = -1;
    }
    __syncthreads();
    const int s_a = i0, s_b = (i0 + pw - 1) % pw;
    int nb_a = 0, nb_b = 0;
    // (eight independent loads asked for before the first is used: one round of memory latency per pass instead of eight at
    // 8192 columns -- the kernel sits on a cold image's critical path, before the job exists)
    for (int c0 = t; c0 < pw; c0 += 8 * 1024) {
        uint32_t te[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + 1024 * k;
            if (EVAL) {
                te[k] = c < pw ? pack_yaw_entry(yaw_row_eval(c, pw, yr)) : 0u;
                if (c < pw)
                    T[c] = te[k];
            } else {
                te[k] = c < pw ? T[c] : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + 1024 * k;
            if (c < pw) {
                nb_a += yaw_delta(te[k], c, s_a, pw) < 0;
                nb_b += yaw_delta(te[k], c, s_b, pw) < 0;
            }
        }
    }
    if (nb_a) atomicAdd(&bad[0], nb_a);
    if (nb_b) atomicAdd(&bad[1], nb_b);
    if (EVAL)
        __threadfence_block();  // (the second pass reads neighbours' entries from the table this workgroup has just written)
    __syncthreads();
    const bool ok_a = bad[0] == 0, ok_b = bad[1] == 0;
    const int s = ok_a ? s_a : s_b;
    if (!ok_a && !ok_b) {
        for (int c = t; c < pw; c += 1024)
            F4[c] = 0u;
        if (t == 0)
            desc[yi] = YawDesc{0, 2, 0, -1};
        return;
    }
    const int c_last = (pw - 1 - s + pw) % pw;  // the rot column whose source column is pw-1
    int lmin = 64, lmax = -1;
    for (int c0 = t; c0 < pw; c0 += 4 * 1024) {
        uint32_t te[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                int cm = c0 + 1024 * k + m;
                if (cm >= pw)
                    cm -= pw;
                te[k][m] = c0 + 1024 * k < pw ? T[cm] : 0u;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + 1024 * k;
            if (c >= pw)
                continue;
            uint32_t w = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                int cm = c + m;
                if (cm >= pw)
                    cm -= pw;
                int d = yaw_delta(te[k][m], cm, s, pw);
                w |= (uint32_t)d << (8 * m);
                if (m == 0 && c != c_last) {
                    lmin = min(lmin, d);
                    lmax = max(lmax, d);
                }
            }
            F4[c] = w;
        }
    }
    atomicMin(&dmin, lmin);
    atomicMax(&dmax, lmax);
    __syncthreads();
    if (t == 0) {
        YawDesc d;
        d.s = s;
        if (dmin == dmax || pw == 1) {
            d.mode = 0;
            d.f = pw == 1 ? 0 : dmin;
            int dl = yaw_delta(T[c_last], c_last, s, pw);
            d.c_clamp = (pw > 1 && dl != d.f) ? c_last : -1;
        } else {
            d.mode = 1;
            d.f = 0;
            d.c_clamp = -1;
        }
        desc[yi] = d;
    }
}

// ---------------------------------------------------------------------------------------------
// pitch map as float32 arrays (get_pitch_mapping drop-in, and the 1e-5 map-parity check)
// ---------------------------------------------------------------------------------------------
__global__ void pitch_map_kernel(float* __restrict__ U, float* __restrict__ V, int ow, int oh,
                                 MapGeom g, float c, float s)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= ow || y >= oh)
        return;
    float uu, vv;
    pitch_map_eval((float)x, (float)y, g, c, s, uu, vv);
    U[(size_t)y * ow + x] = uu;
    V[(size_t)y * ow + x] = vv;
}

// ---------------------------------------------------------------------------------------------
// legacy tool: one combined rotation R = R_pitch @ R_yaw (L:21-45) applied to the normalised pinhole
// ray, then the same spherical mapping (precompute_mapping, L:47-157).  float32 throughout; the 3x3 by
// 3xN sgemm as OpenBLAS accumulates it: acc = R[i][0]*v0; acc = fma(R[i][1], v1, acc); fma(R[i][2], v2, acc)
// (bit-equal to the reference's output for the golden cases).
// ---------------------------------------------------------------------------------------------
struct Rot3 {
    float m[9];
};

__global__ void rot_map_kernel(float* __restrict__ U, float* __restrict__ V, int ow, int oh, MapGeom g, Rot3 R)
{
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y;
    if (px >= ow || py >= oh)
        return;
    const float TWO_PI_F = 6.283185307179586f, PI_F = 3.141592653589793f;
    const float x = (float)px - g.half_w;  // L:107
    const float y = g.half_h - (float)py;  // L:108
    const float z = g.focal;               // L:109
    const float n = __fsqrt_rn(x * x + y * y + z * z);  // L:113
    const float xn = __fdiv_rn(x, n), yn = __fdiv_rn(y, n), zn = __fdiv_rn(z, n);  // L:114-116
    const float xr = __builtin_fmaf(R.m[2], zn, __builtin_fmaf(R.m[1], yn, R.m[0] * xn));  // L:123
    const float yr = __builtin_fmaf(R.m[5], zn, __builtin_fmaf(R.m[4], yn, R.m[3] * xn));
    const float zr = __builtin_fmaf(R.m[8], zn, __builtin_fmaf(R.m[7], yn, R.m[6] * xn));
    const float theta = acosf(zr);  // L:130
    float phi = atan2f(yr, xr);     // L:145, floored '%' of a value in [-pi, pi]
    if (phi < 0.0f)
        phi += TWO_PI_F;
    else if (phi == 0.0f)
        phi = 0.0f;
    float uu = __fdiv_rn(phi * g.pw_f, TWO_PI_F);  // L:157
    float vv = __fdiv_rn(theta * g.ph_f, PI_F);    // L:158
    uu = clip_keep_nan(uu, 0.0f, g.pw_f - 1.0f);   // L:161
    vv = clip_keep_nan(vv, 0.0f, g.ph_f - 1.0f);   // L:162
    U[(size_t)py * ow + px] = uu;
    V[(size_t)py * ow + px] = vv;
}

// ---------------------------------------------------------------------------------------------
// Views whose width is not divisible by 4 have device rows padded to whole 4-pixel groups (ViewsParams::out_row);
// the caller's array is contiguous.  One dword of the contiguous image per thread, its four bytes fetched from
// the padded rows (a row-wise DMA copy instead costs microseconds per row).
// ---------------------------------------------------------------------------------------------
__global__ void compact_rows_kernel(uint32_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n_bytes,
                                    uint32_t row_bytes, uint32_t src_row)
{
    const size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b0 = 4 * d;
    if (b0 >= n_bytes)
        return;
    size_t r = b0 / row_bytes;
    uint32_t c = (uint32_t)(b0 - r * row_bytes);
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (b0 + k < n_bytes)
            w |= (uint32_t)src[r * src_row + c] << (8 * k);
        if (++c == row_bytes) {
            c = 0;
            ++r;
        }
    }
    dst[d] = w;  // (the staging buffer holds whole dwords)
}

// ---------------------------------------------------------------------------------------------
// launchers (called from the p2p_host_*.cpp units through p2p_device.h)
// ---------------------------------------------------------------------------------------------
hipError_t launch_compact_rows(void* dst, const uint8_t* src, size_t n_bytes, int row_bytes, int src_row, hipStream_t st)
{
    const size_t n = (n_bytes + 3) / 4;
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint32_t*)dst, src, n_bytes,
                       (uint32_t)row_bytes, (uint32_t)src_row);
    return hipGetLastError();
}

hipError_t launch_yaw_tables(uint32_t* packed, float* rows, int pw, int n_yaw, const double* yaw_rad,
                             hipStream_t st)
{
    dim3 grid((pw + 255) / 256, n_yaw);
    hipLaunchKernelGGL(yaw_table_kernel, grid, dim3(256), 0, st, packed, rows, pw, yaw_rad);
    return hipGetLastError();
}

hipError_t launch_yaw_pack(uint32_t* packed, const float* rows, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(yaw_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, packed, rows, n);
    return hipGetLastError();
}

hipError_t launch_yaw_desc(YawDesc* desc, uint32_t* f4tab, uint32_t* packed, int pw, int n_yaw, const double* yaw_rad,
                           hipStream_t st)
{
    // yaw_rad != nullptr: the packed table is written by this launch too (no launch_yaw_tables in front of it)
    if (yaw_rad)
        hipLaunchKernelGGL(yaw_desc_kernel<true>, dim3(n_yaw), dim3(1024), 0, st, desc, f4tab, packed, pw, yaw_rad);
    else
        hipLaunchKernelGGL(yaw_desc_kernel<false>, dim3(n_yaw), dim3(1024), 0, st, desc, f4tab, packed, pw, yaw_rad);
    return hipGetLastError();
}

hipError_t launch_pitch_map(float* U, float* V, int ow, int oh, const MapGeom& g, float c, float s,
                            hipStream_t st)
{
    dim3 grid((ow + 255) / 256, oh);
    hipLaunchKernelGGL(pitch_map_kernel, grid, dim3(256), 0, st, U, V, ow, oh, g, c, s);
    return hipGetLastError();
}

hipError_t launch_rot_map(float* U, float* V, int ow, int oh, const MapGeom& g, const float* R9, hipStream_t st)
{
    Rot3 R;
    for (int i = 0; i < 9; ++i)
        R.m[i] = R9[i];
    dim3 grid((ow + 255) / 256, oh);
    hipLaunchKernelGGL(rot_map_kernel, grid, dim3(256), 0, st, U, V, ow, oh, g, R);
    return hipGetLastError();
}

// Robustness self-test (P2P_SCRAMBLE_PLAN, tests/fuzz/scramble_tables.py): overwrite a table with pseudo-random words.
// The view kernels must draw garbage from garbage tables -- and nothing worse.
__global__ void scramble_kernel(uint32_t* __restrict__ p, size_t n_words, uint32_t seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        // a mix of wild values, small values and all-ones, so that every clamp sees both sides
        p[i] = (h & 3u) == 0u ? 0xFFFFFFFFu : ((h & 3u) == 1u ? (h >> 20) : h);
    }
}

hipError_t launch_scramble(void* p, size_t bytes, uint32_t seed, hipStream_t st)
{
    if (!p || bytes < 4)
        return hipSuccess;
    hipLaunchKernelGGL(scramble_kernel, dim3(1024), dim3(256), 0, st, (uint32_t*)p, bytes / 4, seed);
    return hipGetLastError();
}

}  // namespace p2p
