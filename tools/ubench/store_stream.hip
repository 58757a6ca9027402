// Micro-benchmark: the view kernels' STORE pattern alone at sizes that cannot sit in the 256 MB Infinity Cache
// (store_pattern.hip measures config 2, whose 224 MB of views are rewritten in place launch after launch).
//   ./store_stream ow oh n_yaw n_pitch pairs_per_block      e.g. config 4:  4096 4096 72 5 24   (18.1 GB of views)
// Grid (tile, chunk of yaws, pitch view) as remap_views_kernel's; per yaw every lane writes 12 bytes (4 pixels of one
// row) non-temporally through a buffer descriptor.  Also: a plain 16-bytes-per-lane streaming fill of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// RD > 0: every workgroup first reads RD x 4 KB of its own (cold) or of one shared 1 MB window (warm, rd_mask) and
// folds them into what it writes -- the reads of plan tables and source pieces, without their arithmetic
template <int TW, int AUX>
__global__ __launch_bounds__(256) void k(uint8_t* out, int ow, int oh, int n_yaw, int n_pitch, int ppb,
                                         const u32x4* rd = nullptr, int rd_n = 0, size_t rd_mask = ~(size_t)0, int rd_per_yaw = 0)
{
    const int tiles_x = (ow + TW - 1) / TW;
    const int tile = blockIdx.x, chunk = blockIdx.y, pitch = blockIdx.z;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int t = threadIdx.x;
    const size_t view_bytes = (size_t)ow * oh * 3;
    const int lanes_per_row = TW / 4;
    const int passes = (TW * 16 / 4) / 256;
    uint32_t acc = t;
    const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int i = 0; i < rd_n; ++i) {
        const u32x4 v = rd[((wg * rd_n + i) * 256 + t) & rd_mask];
        acc += v.x ^ v.w;
    }
    const int y1 = min(n_yaw, (chunk + 1) * ppb);
    for (int y = chunk * ppb; y < y1; ++y) {
        for (int i = 0; i < rd_per_yaw; ++i) {   // source pieces: the same 4 KB per (tile, pitch) for every yaw, shifted
            const u32x4 v = rd[(((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + i + (size_t)y * 1024) * 256 + t) & rd_mask];
            acc += v.y;
        }
        uint8_t* O = out + ((size_t)y * n_pitch + pitch) * view_bytes;
        for (int p = 0; p < passes; ++p) {
            const int idx = p * 256 + t;
            const int row = ty * 16 + idx / lanes_per_row, col = tx * TW + 4 * (idx % lanes_per_row);
            const bool ok = row < oh && col < ow;
            const u32x3 v = {acc, acc ^ 0x55u, acc + 7u};
            const int off = ok ? (int)(((size_t)row * ow + col) * 3) : 0x7FFFFFFF;
            __builtin_amdgcn_raw_buffer_store_b96(v, __builtin_amdgcn_make_buffer_rsrc(O, 0, (int)view_bytes, 0x00020000), off, 0, AUX);
        }
        acc += 3u;
    }
}

template <int AUX>
__global__ __launch_bounds__(256) void fill(u32x4* out, size_t n16)
{
    const u32x4 v = {1u, 2u, 3u, threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        if (AUX) __builtin_nontemporal_store(v, out + i); else out[i] = v;
}

int main(int argc, char** argv)
{
    const int ow = argc > 1 ? atoi(argv[1]) : 4096, oh = argc > 2 ? atoi(argv[2]) : 4096;
    const int n_yaw = argc > 3 ? atoi(argv[3]) : 72, n_pitch = argc > 4 ? atoi(argv[4]) : 5, ppb = argc > 5 ? atoi(argv[5]) : 24;
    const size_t bytes = (size_t)n_yaw * n_pitch * ow * oh * 3;
    uint8_t* d;
    if (hipMalloc(&d, bytes + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int chunks = (n_yaw + ppb - 1) / ppb;
    const int reps = bytes > (size_t)2e9 ? 6 : 200;
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %9.3f ms per pass, %6.0f GB/s written\n", name, ms / reps, (double)bytes / (ms / reps * 1e-3) / 1e9);
    };
    printf("%d x %d views, %d yaws x %d pitch views, %d yaws per workgroup: %.2f GB\n", ow, oh, n_yaw, n_pitch, ppb, bytes / 1e9);
    const int t64 = ((ow + 63) / 64) * ((oh + 15) / 16), t128 = ((ow + 127) / 128) * ((oh + 15) / 16);
    time("64 x 16 tiles, 12 B per lane, nt", [&] { k<64, 2><<<dim3(t64, chunks, n_pitch), 256>>>(d, ow, oh, n_yaw, n_pitch, ppb); });
    time("64 x 16 tiles, 12 B per lane, default policy", [&] { k<64, 0><<<dim3(t64, chunks, n_pitch), 256>>>(d, ow, oh, n_yaw, n_pitch, ppb); });
    time("128 x 16 tiles, 12 B per lane, nt", [&] { k<128, 2><<<dim3(t128, chunks, n_pitch), 256>>>(d, ow, oh, n_yaw, n_pitch, ppb); });
    u32x4* rd; const size_t rd_bytes = (size_t)4 << 30;   // 4 GB to read from
    if (hipMalloc(&rd, rd_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(rd, 1, rd_bytes);
    const size_t cold = rd_bytes / 16 - 1, warm = ((size_t)1 << 20) / 16 - 1;
    time("64 x 16 nt + 8 KB cold reads per workgroup", [&] { k<64, 2><<<dim3(t64, chunks, n_pitch), 256>>>(d, ow, oh, n_yaw, n_pitch, ppb, rd, 2, cold); });
    time("64 x 16 nt + 8 KB warm reads per workgroup", [&] { k<64, 2><<<dim3(t64, chunks, n_pitch), 256>>>(d, ow, oh, n_yaw, n_pitch, ppb, rd, 2, warm); });
    time("64 x 16 nt + 8 KB cold + 4 KB per yaw (cold-ish)", [&] { k<64, 2><<<dim3(t64, chunks, n_pitch), 256>>>(d, ow, oh, n_yaw, n_pitch, ppb, rd, 2, cold, 1); });
    time("64 x 16 nt + 8 KB warm + 4 KB per yaw (warm)", [&] { k<64, 2><<<dim3(t64, chunks, n_pitch), 256>>>(d, ow, oh, n_yaw, n_pitch, ppb, rd, 2, warm, 1); });
    time("streaming fill, 16 B per lane, nt", [&] { fill<1><<<dim3(256 * 16), 256>>>((u32x4*)d, bytes / 16); });
    time("streaming fill, 16 B per lane, default policy", [&] { fill<0><<<dim3(256 * 16), 256>>>((u32x4*)d, bytes / 16); });
    return 0;
}
