// p2p_kernels.hip -- gfx950 (MI355X / CDNA4) device code for the equirectangular ->
// perspective view-synthesis hot path.  Reference behaviour (cited, never copied):
//   P = /root/reference/app/panorama_to_plane-pitch.py
//     yaw map            P:79-108      -> yaw_table_kernel (bit-exact dtype flow: f32, then f64)
//     pitch map          P:114-175     -> pitch_map_eval (f32), pitch_map_kernel
//     cv2.remap x2       P:192-199,212-218 -> remap_views_kernel (both stages fused, fixed point)
//   L = /root/reference/app/legacy/panorama_to_plane.py
//     panorama_to_plane  L:159-194     -> remap_maps_kernel (generic cv2.remap INTER_LINEAR, u8)
// The fixed-point arithmetic is OpenCV 4.10's (imgwarp.cpp remapBilinear, INTER_BITS = 5,
// INTER_REMAP_COEF_BITS = 15); see DESIGN.md "Arithmetic contract".
//
// Compiled with -ffp-contract=off: every float operation below rounds where NumPy rounds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "p2p_device.h"

namespace p2p {

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_round_f32(float v)
{
    // cvRound(float) on x86-64 = cvtss2si: round-half-even, NaN / out of range -> INT_MIN
    if (!(v >= -2147483648.0f && v < 2147483648.0f))
        return INT32_MIN;
    return (int)__builtin_rintf(v);
}

__device__ __forceinline__ int sat_short(int v)
{
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

__device__ __forceinline__ float clip_keep_nan(float v, float lo, float hi)
{
    // np.clip propagates NaN; fminf/fmaxf would not
    return v < lo ? lo : (v > hi ? hi : v);
}

// P:114-175 for one output pixel, float32 throughout, same operation order as NumPy:
//   x**2 + y**2 + z**2 left to right, IEEE sqrt and divide, the 3x3 float32 sgemm as the
//   sequential-FMA accumulation OpenBLAS performs (acc = fma(R[i][k], v[k], acc), k = 0..2),
//   arccos / arctan2 % 2pi, scale, clip.
__device__ __forceinline__ void pitch_map_eval(float u, float v, const MapGeom& g, float c, float s,
                                               float& U, float& V)
{
    const float TWO_PI_F = 6.283185307179586f;  // float32(2*np.pi), the "weak" Python scalar
    const float PI_F = 3.141592653589793f;
    float x = u - g.half_w;  // P:129
    float y = g.half_h - v;  // P:130
    float z = g.focal;       // P:131
    float n = __fsqrt_rn(x * x + y * y + z * z);  // P:134
    float xn = __fdiv_rn(x, n), yn = __fdiv_rn(y, n), zn = __fdiv_rn(z, n);  // P:137-139
    float yr = __builtin_fmaf(-s, zn, c * yn);  // P:155 row 1: [0, cos, -sin]
    float zr = __builtin_fmaf(c, zn, s * yn);   // P:155 row 2: [0, sin,  cos]
    float theta = acosf(zr);                    // P:162 (NaN if zr rounds above 1)
    float phi = atan2f(yr, xn);                 // P:164
    if (phi < 0.0f)
        phi += TWO_PI_F;  // floored '%': |phi| <= pi so fmod is the identity; -0.0 -> +0.0 either way
    else if (phi == 0.0f)
        phi = 0.0f;
    U = __fdiv_rn(phi * g.pw_f, TWO_PI_F);  // P:167
    V = __fdiv_rn(theta * g.ph_f, PI_F);    // P:169
    U = clip_keep_nan(U, 0.0f, g.pw_f - 1.0f);  // P:172
    V = clip_keep_nan(V, 0.0f, g.ph_f - 1.0f);  // P:173
}

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
    double m = fmod(pr, TWO_PI_D);                    // NumPy's floored '%'
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
// Stage 1 for one pixel of the yaw-resampled panorama ("rot"), P:192-199:
//   rot[r][c] = ((32-f)*src[r][i] + f*src[r][i+1] + 16) >> 5  per channel, (i, f) = yaw table[c]
// (cv::remap with fy == 0: weights 1024*(32-f), 1024*f, rounding 1<<14, shift 15).
// Returns the pixel as a dword B | G<<8 | R<<16.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rot_pixel(const uint8_t* __restrict__ row, uint32_t te)
{
    uint32_t off = te & 0xFFFFFu;
    uint32_t f = te >> 20;
    uint2 q;
    __builtin_memcpy(&q, row + off, 8);  // unaligned 8-byte load: pixels i and i+1 (6 bytes used)
    uint32_t p0 = q.x;
    uint32_t p1 = __builtin_amdgcn_alignbyte(q.y, q.x, 3);
    uint32_t g = 32u - f;
    uint32_t br = g * (p0 & 0x00FF00FFu) + f * (p1 & 0x00FF00FFu) + 0x00100010u;
    uint32_t gg = g * (p0 & 0x0000FF00u) + f * (p1 & 0x0000FF00u) + 0x00001000u;
    return ((br >> 5) & 0x00FF00FFu) | ((gg >> 5) & 0x0000FF00u);
}

// Stage 2 for one output pixel, P:212-218: bilinear blend of four rot pixels with cv::remap's
// weights 32*(32-fx)(32-fy).. and (sum + 16384) >> 15  ==  (sum' + 512) >> 10 with weights / 32.
__device__ __forceinline__ uint32_t blend4(uint32_t a, uint32_t b, uint32_t c, uint32_t d,
                                           uint32_t fx, uint32_t fy)
{
    uint32_t gx = 32u - fx, gy = 32u - fy;
    // horizontal: two channels per 32-bit op (16-bit fields hold <= 32*255)
    uint32_t h0br = gx * (a & 0x00FF00FFu) + fx * (b & 0x00FF00FFu);
    uint32_t h1br = gx * (c & 0x00FF00FFu) + fx * (d & 0x00FF00FFu);
    uint32_t h0g = gx * ((a >> 8) & 0xFFu) + fx * ((b >> 8) & 0xFFu);
    uint32_t h1g = gx * ((c >> 8) & 0xFFu) + fx * ((d >> 8) & 0xFFu);
    uint32_t vb = (gy * (h0br & 0xFFFFu) + fy * (h1br & 0xFFFFu) + 512u) >> 10;
    uint32_t vr = (gy * (h0br >> 16) + fy * (h1br >> 16) + 512u) >> 10;
    uint32_t vg = (gy * h0g + fy * h1g + 512u) >> 10;
    return vb | (vg << 8) | (vr << 16);
}

// ---------------------------------------------------------------------------------------------
// The hot kernel: one workgroup = one TILE_W x TILE_H tile of output pixels of one pitch view;
// it keeps the pitch-stage coordinates of its pixels in registers and loops over a chunk of
// (panorama, yaw) pairs.  Per pair it materialises the tile's footprint of the yaw-resampled
// panorama in LDS (stage 1, exact uint8 intermediate), then gathers the 2x2 taps from LDS
// (stage 2).  Footprints too large for LDS (views containing a pole) gather straight from
// global memory with the same arithmetic.
// ---------------------------------------------------------------------------------------------
template <bool HOST_MAPS>
__global__ __launch_bounds__(VIEWS_BLOCK) void remap_views_kernel(ViewsParams P)
{
    __shared__ uint32_t tile[2][LDS_TILE_CAP];
    __shared__ int bbox[4];

    const int t = threadIdx.x;
    const int tiles_x = (P.ow + TILE_W - 1) / TILE_W;
    const int tile_id = blockIdx.x;
    const int pitch_i = blockIdx.y;
    const int x0 = (tile_id % tiles_x) * TILE_W;
    const int y0 = (tile_id / tiles_x) * TILE_H;
    const int px = x0 + (t % TILE_W);
    const int py = y0 + (t / TILE_W);
    const bool inside = px < P.ow && py < P.oh;

    // ---- pitch-stage coordinate of this thread's pixel, quantised as cv::remap does ----
    int sx = INT32_MIN, sy = INT32_MIN;
    if (inside) {
        float U, V;
        if (HOST_MAPS) {
            size_t k = ((size_t)pitch_i * P.oh + py) * P.ow + px;
            U = P.mapU[k];
            V = P.mapV[k];
        } else {
            PitchConst pc = P.pitch[pitch_i];
            pitch_map_eval((float)px, (float)py, P.geom, pc.c, pc.s, U, V);
        }
        sx = cv_round_f32(U * 32.0f);
        sy = cv_round_f32(V * 32.0f);
        if (P.coords && blockIdx.z == 0) {
            size_t k = (((size_t)pitch_i * P.oh + py) * P.ow + px) * 2;
            P.coords[k] = sx;
            P.coords[k + 1] = sy;
        }
    }
    int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);
    const uint32_t fx = (uint32_t)sx & 31u, fy = (uint32_t)sy & 31u;
    // A pixel contributes only if its 2x2 footprint touches the panorama (BORDER_CONSTANT 0:
    // cv::remap writes borderValue when sx >= w || sx+1 < 0 || sy >= h || sy+1 < 0).  For the
    // reference's clipped maps that is every pixel except NaN ones (ix = iy = -32768).
    const bool live = inside && ix >= -1 && iy >= -1 && ix < P.pw && iy < P.ph;

    // ---- footprint of the tile in rot space ----
    if (t < 4)
        bbox[t] = (t & 1) ? -2 : INT32_MAX;  // [0]=min x, [1]=max x, [2]=min y, [3]=max y
    __syncthreads();
    if (live) {
        atomicMin(&bbox[0], ix);
        atomicMax(&bbox[1], ix);
        atomicMin(&bbox[2], iy);
        atomicMax(&bbox[3], iy);
    }
    __syncthreads();
    const int c0 = bbox[0], c1 = bbox[1], r0 = bbox[2], r1 = bbox[3];
    const bool any_live = c1 >= -1;
    const int Wt = c1 - c0 + 2, Ht = r1 - r0 + 2;  // +1 for the right / lower taps
    const int area = Wt * Ht;
    const bool use_lds = any_live && Wt <= 255 && area <= LDS_TILE_CAP;
    // idx / Wt by multiply-shift: exact for idx*Wt < 2^20 (idx < 4096, Wt < 256)
    const uint32_t magic = use_lds ? ((1u << 20) + (uint32_t)Wt - 1u) / (uint32_t)Wt : 0u;
    const int tap = live ? (iy - r0) * Wt + (ix - c0) : 0;

    // output addressing: 4 horizontally adjacent pixels = 12 bytes = 3 aligned dwords
    const int lane4 = t & 3;
    const bool fast_store = (P.ow & 3) == 0;
    const size_t view_bytes = (size_t)P.oh * P.ow * 3;

    const int pair0 = blockIdx.z * P.pairs_per_block;
    int pair1 = pair0 + P.pairs_per_block;
    const int n_pairs = P.n_panos * P.n_yaw;
    if (pair1 > n_pairs)
        pair1 = n_pairs;

    int buf = 0;
    for (int pair = pair0; pair < pair1; ++pair, buf ^= 1) {
        const int pano_i = pair / P.n_yaw;
        const int yaw_i = pair - pano_i * P.n_yaw;
        const uint8_t* __restrict__ S = P.src + (size_t)pano_i * P.pano_stride;
        const uint32_t* __restrict__ T = P.ytab + (size_t)yaw_i * P.pw;
        uint32_t pix = 0;

        if (use_lds) {
            uint32_t* tl = tile[buf];
            for (int idx = t; idx < area; idx += VIEWS_BLOCK) {
                int rr = (int)(((uint32_t)idx * magic) >> 20);
                int cc = idx - rr * Wt;
                int r = r0 + rr, c = c0 + cc;
                uint32_t val = 0;
                if (r >= 0 && c >= 0 && r < P.ph && c < P.pw)
                    val = rot_pixel(S + (size_t)r * P.src_pitch, T[c]);
                tl[idx] = val;
            }
            __syncthreads();
            if (live)
                pix = blend4(tl[tap], tl[tap + 1], tl[tap + Wt], tl[tap + Wt + 1], fx, fy);
        } else if (live) {
            // direct gather (pole-containing footprints): same arithmetic, taps from global memory
            const bool c0in = ix >= 0, c1in = ix + 1 < P.pw, r0in = iy >= 0, r1in = iy + 1 < P.ph;
            const uint8_t* row0 = S + (ptrdiff_t)iy * P.src_pitch;
            const uint8_t* row1 = row0 + P.src_pitch;
            const uint32_t t0 = c0in ? T[ix] : 0u, t1 = c1in ? T[ix + 1] : 0u;
            uint32_t a = (c0in && r0in) ? rot_pixel(row0, t0) : 0u;
            uint32_t b = (c1in && r0in) ? rot_pixel(row0, t1) : 0u;
            uint32_t c = (c0in && r1in) ? rot_pixel(row1, t0) : 0u;
            uint32_t d = (c1in && r1in) ? rot_pixel(row1, t1) : 0u;
            pix = blend4(a, b, c, d, fx, fy);
        }

        // ---- store: [pano][yaw][pitch][oh][ow][3] ----
        uint8_t* O = P.out + ((size_t)pair * P.n_pitch + pitch_i) * view_bytes;
        if (fast_store) {
            // lanes 4k..4k+3 hold pixels P0..P3; lanes with lane4 < 3 emit dword lane4 of the 12 bytes
            uint32_t nxt = __shfl_down(pix, 1);
            uint32_t dw = __builtin_amdgcn_alignbit(nxt, pix << 8, 8u * (uint32_t)(lane4 + 1));
            if (inside && lane4 < 3)
                *reinterpret_cast<uint32_t*>(O + ((size_t)py * P.ow + px) * 3 + lane4) = dw;
        } else if (inside) {
            uint8_t* o = O + ((size_t)py * P.ow + px) * 3;
            o[0] = (uint8_t)pix;
            o[1] = (uint8_t)(pix >> 8);
            o[2] = (uint8_t)(pix >> 16);
        }
        // the other LDS buffer is written next iteration; its readers finished before the
        // barrier above, so one barrier per pair suffices
    }
}

// ---------------------------------------------------------------------------------------------
// Generic cv2.remap(src, U, V, INTER_LINEAR, border) for uint8, cn in {1,3,4}
// (panorama_to_plane, L:159-194).  One thread per destination pixel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int border_interpolate(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (border == 1)  // REPLICATE
        return p < 0 ? 0 : len - 1;
    if (border == 2 || border == 4) {  // REFLECT / REFLECT_101
        int delta = border == 4;
        if (len == 1)
            return 0;
        do {
            if (p < 0)
                p = -p - 1 + delta;
            else
                p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (border == 3) {  // WRAP
        if (p < 0)
            p -= ((p - len + 1) / len) * len;
        if (p >= len)
            p %= len;
        return p;
    }
    return -1;  // CONSTANT
}

template <int CN>
__global__ void remap_maps_kernel(RemapParams P)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= P.ow || y >= P.oh)
        return;
    size_t k = (size_t)y * P.ow + x;
    int qx = cv_round_f32(P.U[k] * 32.0f);
    int qy = cv_round_f32(P.V[k] * 32.0f);
    int sx = sat_short(qx >> 5), sy = sat_short(qy >> 5);
    int fx = qx & 31, fy = qy & 31;
    int w0 = (32 - fx) * (32 - fy), w1 = fx * (32 - fy), w2 = (32 - fx) * fy, w3 = fx * fy;
    uint8_t* D = P.dst + k * CN;
    const uint8_t* cval = P.cval;
    if (P.border == 0 && (sx >= P.sw || sx + 1 < 0 || sy >= P.sh || sy + 1 < 0)) {
#pragma unroll
        for (int ch = 0; ch < CN; ++ch)
            D[ch] = cval[ch];
        return;
    }
    int sx0, sx1, sy0, sy1;
    if (P.border == 1) {
        sx0 = min(max(sx, 0), P.sw - 1);
        sx1 = min(max(sx + 1, 0), P.sw - 1);
        sy0 = min(max(sy, 0), P.sh - 1);
        sy1 = min(max(sy + 1, 0), P.sh - 1);
    } else {
        sx0 = border_interpolate(sx, P.sw, P.border);
        sx1 = border_interpolate(sx + 1, P.sw, P.border);
        sy0 = border_interpolate(sy, P.sh, P.border);
        sy1 = border_interpolate(sy + 1, P.sh, P.border);
    }
    const uint8_t* v0 = (sx0 >= 0 && sy0 >= 0) ? P.src + (size_t)sy0 * P.src_pitch + (size_t)sx0 * CN : nullptr;
    const uint8_t* v1 = (sx1 >= 0 && sy0 >= 0) ? P.src + (size_t)sy0 * P.src_pitch + (size_t)sx1 * CN : nullptr;
    const uint8_t* v2 = (sx0 >= 0 && sy1 >= 0) ? P.src + (size_t)sy1 * P.src_pitch + (size_t)sx0 * CN : nullptr;
    const uint8_t* v3 = (sx1 >= 0 && sy1 >= 0) ? P.src + (size_t)sy1 * P.src_pitch + (size_t)sx1 * CN : nullptr;
#pragma unroll
    for (int ch = 0; ch < CN; ++ch) {
        int a = v0 ? v0[ch] : cval[ch];
        int b = v1 ? v1[ch] : cval[ch];
        int c = v2 ? v2[ch] : cval[ch];
        int d = v3 ? v3[ch] : cval[ch];
        // (32*sum + 16384) >> 15 == (sum + 512) >> 10; the table's {32767,0,0,1} cell for
        // fx == fy == 0 yields the same byte (|p11 - p00| < 16384), see tests/test_oracle_remap.py
        D[ch] = (uint8_t)((w0 * a + w1 * b + w2 * c + w3 * d + 512) >> 10);
    }
}

// ---------------------------------------------------------------------------------------------
// launchers (called from p2p_host.cpp through p2p_device.h)
// ---------------------------------------------------------------------------------------------
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

hipError_t launch_pitch_map(float* U, float* V, int ow, int oh, const MapGeom& g, float c, float s,
                            hipStream_t st)
{
    dim3 grid((ow + 255) / 256, oh);
    hipLaunchKernelGGL(pitch_map_kernel, grid, dim3(256), 0, st, U, V, ow, oh, g, c, s);
    return hipGetLastError();
}

hipError_t launch_remap_views(const ViewsParams& P, bool host_maps, hipStream_t st)
{
    const int tiles = ((P.ow + TILE_W - 1) / TILE_W) * ((P.oh + TILE_H - 1) / TILE_H);
    const int n_pairs = P.n_panos * P.n_yaw;
    const int zblocks = (n_pairs + P.pairs_per_block - 1) / P.pairs_per_block;
    dim3 grid(tiles, P.n_pitch, zblocks);
    if (host_maps)
        hipLaunchKernelGGL(remap_views_kernel<true>, grid, dim3(VIEWS_BLOCK), 0, st, P);
    else
        hipLaunchKernelGGL(remap_views_kernel<false>, grid, dim3(VIEWS_BLOCK), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_remap_maps(const RemapParams& P, int cn, hipStream_t st)
{
    dim3 block(64, 4);
    dim3 grid((P.ow + 63) / 64, (P.oh + 3) / 4);
    if (cn == 1)
        hipLaunchKernelGGL(remap_maps_kernel<1>, grid, block, 0, st, P);
    else if (cn == 3)
        hipLaunchKernelGGL(remap_maps_kernel<3>, grid, block, 0, st, P);
    else
        hipLaunchKernelGGL(remap_maps_kernel<4>, grid, block, 0, st, P);
    return hipGetLastError();
}

}  // namespace p2p
