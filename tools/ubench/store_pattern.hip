// Micro-benchmark: what the view kernels' STORE pattern alone reaches on HBM (gfx950).  Config 2's geometry: 36 views
// of 1920 x 1080 x 3 bytes; one workgroup per 64 x 16 tile and pitch view loops over 12 yaws and, per yaw, every lane
// writes 12 bytes (4 pixels) non-temporally -- a wave covers 4 rows x 192 bytes, exactly as remap_views_kernel does --
// with nothing else going on.  Variants: 128-wide tiles (a wave covers 2 rows x 384 bytes = whole 128-byte lines),
// (`spin`: optional VALU work per yaw).  Measured: 47.3 us = 4.7 TB/s and 42.6 us = 5.3 TB/s -- the view kernel's
// writes (2.55 TB/s over its 88 us) are at half of what their pattern could reach: not its limit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

template <int TW>  // tile width: 64 (16 rows) or 128 (8 rows per 256-thread workgroup pass, 2 passes)
__global__ __launch_bounds__(256) void k(uint8_t* out, int ow, int oh, int n_yaw, int n_pitch, int spin)
{
    const int tiles_x = (ow + TW - 1) / TW, tiles_y = (oh + 15) / 16;
    const int tile = blockIdx.x, pitch = blockIdx.y;
    if (tile >= tiles_x * tiles_y) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int t = threadIdx.x;
    const size_t view_bytes = (size_t)ow * oh * 3;
    const int lanes_per_row = TW / 4;                       // 4 pixels per lane
    const int passes = (TW * 16 / 4) / 256;                 // 1 for 64-wide, 2 for 128-wide
    uint32_t acc = t;
    for (int y = 0; y < n_yaw; ++y) {
        for (int s = 0; s < spin; ++s) acc = acc * 1664525u + 1013904223u;
        uint8_t* O = out + ((size_t)y * n_pitch + pitch) * view_bytes;
        for (int p = 0; p < passes; ++p) {
            const int idx = p * 256 + t;
            const int row = ty * 16 + idx / lanes_per_row, col = tx * TW + 4 * (idx % lanes_per_row);
            const bool ok = row < oh && col < ow;
            const u32x3 v = {acc, acc ^ 0x55u, acc + 7u};
            const int off = ok ? (int)(((size_t)row * ow + col) * 3) : 0x7FFFFFFF;
            __builtin_amdgcn_raw_buffer_store_b96(v, __builtin_amdgcn_make_buffer_rsrc(O, 0, (int)view_bytes, 0x00020000), off, 0, 2);
        }
    }
}

template <int TW>
void run(const char* name, uint8_t* d, int spin)
{
    const int ow = 1920, oh = 1080, n_yaw = 12, n_pitch = 3;
    const int tiles = ((ow + TW - 1) / TW) * ((oh + 15) / 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 300; ++i) k<TW><<<dim3(tiles, n_pitch), 256>>>(d, ow, oh, n_yaw, n_pitch, spin);
    hipEventRecord(e0);
    const int n = 1000;
    for (int i = 0; i < n; ++i) k<TW><<<dim3(tiles, n_pitch), 256>>>(d, ow, oh, n_yaw, n_pitch, spin);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 36.0 * ow * oh * 3;
    printf("%-40s spin %4d: %7.1f us per launch, %6.0f GB/s written\n", name, spin, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e9);
}

int main()
{
    uint8_t* d; hipMalloc(&d, (size_t)36 * 1920 * 1080 * 3 + 4096);
    run<64>("64 x 16 tiles (wave: 4 rows x 192 B)", d, 0);
    run<128>("128 x 16 tiles (wave: 2 rows x 384 B)", d, 0);
    return 0;
}
