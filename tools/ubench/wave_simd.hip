// Micro-benchmark: which SIMD of its CU does wave w of a 256-thread workgroup run on?  (HW_ID register, gfx9 layout:
// wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh 12, se 15:13 ...)  Prints the histogram of (wave index -> SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k(uint32_t* out, int spin)
{
    __shared__ uint4 pad[1600];  // 25.6 KB of LDS, as the view kernel: six workgroups per CU
    uint32_t hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    uint32_t acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1664525u + 1013904223u;
    pad[threadIdx.x].x = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (hw & 0xFFFFu) | (pad[(threadIdx.x + 1) & 255].x & 0u);
}
int main()
{
    const int n = 20000;
    uint32_t* d; hipMalloc(&d, n * 16);
    k<<<n, 256>>>(d, 2000);
    hipDeviceSynchronize();
    static uint32_t h[n * 4];
    hipMemcpy(h, d, n * 16, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    int same_cu = 0;
    for (int b = 0; b < n; ++b) {
        for (int w = 0; w < 4; ++w) hist[w][(h[b * 4 + w] >> 4) & 3]++;
        same_cu += ((h[b * 4] >> 8) & 0xFF) == ((h[b * 4 + 3] >> 8) & 0xFF);
    }
    for (int w = 0; w < 4; ++w)
        printf("wave %d of a workgroup -> SIMD 0..3: %6d %6d %6d %6d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("first 6 workgroups (simd of waves 0..3): ");
    for (int b = 0; b < 6; ++b) printf("[%u %u %u %u] ", (h[b*4]>>4)&3, (h[b*4+1]>>4)&3, (h[b*4+2]>>4)&3, (h[b*4+3]>>4)&3);
    printf("\n");
    return 0;
}
