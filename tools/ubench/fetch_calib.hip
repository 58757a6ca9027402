// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access pattern of remap_views_kernel's
// stage 1: every lane loads a 4-byte-aligned 16-byte piece, neighbouring lanes 12 bytes apart.
// Kernel `pieces` reads a buffer of known size exactly once that way; kernel `stream16` reads it as
// plain 16-byte-aligned dwordx4; both reduce to one dword per block so that nothing is written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((aligned(4))) Q16 { uint32_t d[4]; };
__global__ __launch_bounds__(256) void pieces(const uint8_t* __restrict__ src, size_t n_pieces, uint32_t* out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_pieces; i += (size_t)gridDim.x * 256) {
        Q16 q = *reinterpret_cast<const Q16*>(src + 12 * i);
        acc ^= q.d[0] ^ q.d[1] ^ q.d[2] ^ q.d[3];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void stream16(const uint4* __restrict__ src, size_t n, uint32_t* out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 q = src[i];
        acc ^= q.x ^ q.y ^ q.z ^ q.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
int main()
{
    const size_t bytes = 768ull << 20;  // > 256 MiB Infinity Cache
    uint8_t* d; uint32_t* o;
    hipMalloc(&d, bytes + 64); hipMalloc(&o, 4096 * 4);
    hipMemset(d, 1, bytes + 64);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        pieces<<<2048, 256>>>(d, bytes / 12, o);
        stream16<<<2048, 256>>>((const uint4*)d, bytes / 16, o);
    }
    hipDeviceSynchronize();
    printf("read %zu bytes per kernel launch\n", bytes);
    return 0;
}
