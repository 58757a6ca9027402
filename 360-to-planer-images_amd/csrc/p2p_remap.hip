// p2p_remap.hip -- generic cv2.remap for uint8 (INTER_LINEAR, INTER_NEAREST, INTER_CUBIC; 1 / 3 / 4 channels; five border modes)
//   interpolate_color  L:159-180  -> remap_maps_kernel (INTER_LINEAR), remap_maps_nearest_kernel,
//                                    remap_maps_cubic_kernel + cubic_tab_kernel
// Reference behaviour (cited, never copied):
//   P = /root/reference/app/panorama_to_plane-pitch.py, L = /root/reference/app/legacy/panorama_to_plane.py
// The fixed-point arithmetic is OpenCV 4.10's (imgwarp.cpp remapBilinear, INTER_BITS = 5,
// INTER_REMAP_COEF_BITS = 15); see DESIGN.md "Arithmetic contract".
// Compiled with -ffp-contract=off: every float operation below rounds where NumPy rounds.
#include "p2p_inline.h"

namespace p2p {

// ---------------------------------------------------------------------------------------------
// Generic cv2.remap(src, U, V, INTER_LINEAR, border) for uint8, cn in {1,3,4}
// (panorama_to_plane, L:159-194).  One thread per destination pixel.
// ---------------------------------------------------------------------------------------------
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
// The other two rows of the legacy tool's method table (L:172-176): INTER_NEAREST and INTER_CUBIC, as
// OpenCV 4.10 evaluates them for uint8 (remapNearest; remapBicubic with the 15-bit fixed-point table).
// ---------------------------------------------------------------------------------------------
template <int CN>
__global__ void remap_maps_nearest_kernel(RemapParams P)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= P.ow || y >= P.oh)
        return;
    const size_t k = (size_t)y * P.ow + x;
    // saturate_cast<short>(float): cvRound (half-even), then saturation
    int sx = sat_short(cv_round_f32(P.U[k])), sy = sat_short(cv_round_f32(P.V[k]));
    uint8_t* D = P.dst + k * CN;
    const uint8_t* S = nullptr;
    if ((unsigned)sx < (unsigned)P.sw && (unsigned)sy < (unsigned)P.sh) {
        S = P.src + (size_t)sy * P.src_pitch + (size_t)sx * CN;
    } else if (P.border != 0) {
        sx = border_interpolate(sx, P.sw, P.border);
        sy = border_interpolate(sy, P.sh, P.border);
        S = P.src + (size_t)sy * P.src_pitch + (size_t)sx * CN;
    }
#pragma unroll
    for (int ch = 0; ch < CN; ++ch)
        D[ch] = S ? S[ch] : P.cval[ch];
}

// initInterTab2D(INTER_CUBIC, fixpt): thread (fy, fx) builds its 4x4 cell of shorts.  interpolateCubic with
// A = -0.75 in float32, products scaled by 32768 and rounded half-even, then the cell sum forced to 32768 by
// adjusting the largest (or smallest) of the four entries [2..3][2..3] -- the window OpenCV scans.
__device__ __forceinline__ void cubic_coeffs(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

__global__ void cubic_tab_kernel(short* __restrict__ tab)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= 1024)
        return;
    const float scale = 1.f / 32;
    float cy[4], cx[4];
    cubic_coeffs((cell >> 5) * scale, cy);
    cubic_coeffs((cell & 31) * scale, cx);
    short w[16];
    int isum = 0;
    for (int k1 = 0; k1 < 4; ++k1)
        for (int k2 = 0; k2 < 4; ++k2) {
            const float v = cy[k1] * cx[k2];
            w[k1 * 4 + k2] = (short)sat_short(cv_round_f32(v * 32768.0f));
            isum += w[k1 * 4 + k2];
        }
    if (isum != 32768) {
        const int diff = isum - 32768;
        int M = 2 * 4 + 2, m = 2 * 4 + 2;
        for (int k1 = 2; k1 < 4; ++k1)
            for (int k2 = 2; k2 < 4; ++k2) {
                const int i = k1 * 4 + k2;
                if (w[i] < w[m])
                    m = i;
                else if (w[i] > w[M])
                    M = i;
            }
        if (diff < 0)
            w[M] = (short)(w[M] - diff);
        else
            w[m] = (short)(w[m] - diff);
    }
    for (int i = 0; i < 16; ++i)
        tab[cell * 16 + i] = w[i];
}

template <int CN>
__global__ void remap_maps_cubic_kernel(RemapParams P)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= P.ow || y >= P.oh)
        return;
    const size_t k = (size_t)y * P.ow + x;
    const int qx = cv_round_f32(P.U[k] * 32.0f), qy = cv_round_f32(P.V[k] * 32.0f);
    const int sx = sat_short(qx >> 5) - 1, sy = sat_short(qy >> 5) - 1;
    const short* __restrict__ w = P.ctab + ((qy & 31) * 32 + (qx & 31)) * 16;
    uint8_t* D = P.dst + k * CN;
    if (P.border == 0 && (sx >= P.sw || sx + 4 <= 0 || sy >= P.sh || sy + 4 <= 0)) {
#pragma unroll
        for (int ch = 0; ch < CN; ++ch)
            D[ch] = P.cval[ch];
        return;
    }
    int xs[4], ys[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xs[i] = border_interpolate(sx + i, P.sw, P.border);
        ys[i] = border_interpolate(sy + i, P.sh, P.border);
    }
#pragma unroll
    for (int ch = 0; ch < CN; ++ch) {
        // sum = cval * 32768 + sum (p - cval) * w over the taps that exist == sum p * w with cval at the missing
        // taps, because the 16 weights add up to 32768
        const int cv = P.cval[ch];
        int sum = cv << 15;
        for (int r = 0; r < 4; ++r) {
            if (ys[r] < 0)
                continue;
            const uint8_t* S = P.src + (size_t)ys[r] * P.src_pitch;
            for (int c = 0; c < 4; ++c)
                if (xs[c] >= 0)
                    sum += ((int)S[xs[c] * CN + ch] - cv) * (int)w[r * 4 + c];
        }
        sum = (sum + 16384) >> 15;
        D[ch] = (uint8_t)(sum < 0 ? 0 : (sum > 255 ? 255 : sum));
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_cubic_tab(short* tab, hipStream_t st)
{
    hipLaunchKernelGGL(cubic_tab_kernel, dim3(4), dim3(256), 0, st, tab);
    return hipGetLastError();
}

hipError_t launch_remap_maps(const RemapParams& P, int cn, int interpolation, hipStream_t st)
{
    dim3 block(64, 4);
    dim3 grid((P.ow + 63) / 64, (P.oh + 3) / 4);
    if (interpolation == 0) {
        if (cn == 1)
            hipLaunchKernelGGL(remap_maps_nearest_kernel<1>, grid, block, 0, st, P);
        else if (cn == 3)
            hipLaunchKernelGGL(remap_maps_nearest_kernel<3>, grid, block, 0, st, P);
        else
            hipLaunchKernelGGL(remap_maps_nearest_kernel<4>, grid, block, 0, st, P);
        return hipGetLastError();
    }
    if (interpolation == 2) {
        if (cn == 1)
            hipLaunchKernelGGL(remap_maps_cubic_kernel<1>, grid, block, 0, st, P);
        else if (cn == 3)
            hipLaunchKernelGGL(remap_maps_cubic_kernel<3>, grid, block, 0, st, P);
        else
            hipLaunchKernelGGL(remap_maps_cubic_kernel<4>, grid, block, 0, st, P);
        return hipGetLastError();
    }
    if (cn == 1)
        hipLaunchKernelGGL(remap_maps_kernel<1>, grid, block, 0, st, P);
    else if (cn == 3)
        hipLaunchKernelGGL(remap_maps_kernel<3>, grid, block, 0, st, P);
    else
        hipLaunchKernelGGL(remap_maps_kernel<4>, grid, block, 0, st, P);
    return hipGetLastError();
}

}  // namespace p2p
