// Micro-benchmark + correctness probe: ds_read_b64 from 4-byte-aligned (not 8-byte-aligned) LDS addresses on gfx950.
// The view kernel's stage 2 reads, per output pixel and row, two horizontally adjacent rot pixels = two adjacent dwords
// at an arbitrary dword offset.  Today that is a ds_read2_b32 (32 banks: 32 neighbouring pixels span 33-36 dwords, so
// nearly every wave instruction pays a 2-way conflict); a ds_read_b64 is banked over 64 dwords.  Does the hardware
// accept the misaligned address (SH_MEM_CONFIG alignment mode), does it return the right bytes, and what does it cost?
//   ./lds_b64            prints mismatches and ns per wave-instruction for both forms over four address patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int LDS_DW = 8192;  // 32 KB

// MODE 0: ds_read2_b32 offset1:1   1: ds_read_b64 (inline asm: the compiler would not emit it for align 4)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, const uint32_t* addr_dw, int iters, int check)
{
    __shared__ uint32_t lds[LDS_DW];
    for (int i = threadIdx.x; i < LDS_DW; i += 256)
        lds[i] = 0x9E3779B9u * (uint32_t)i + 12345u;
    __syncthreads();
    uint32_t a[8];
    for (int j = 0; j < 8; ++j)
        a[j] = (uint32_t)(uintptr_t)lds + 4u * addr_dw[j * 256 + threadIdx.x];  // LDS byte addresses
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t v[8];
        const uint32_t step = 8u * (uint32_t)(it & 7);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t ad = a[j] + step;
            if (MODE == 0)
                asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v[j]) : "v"(ad));
            else
                asm volatile("ds_read_b64 %0, %1" : "=v"(v[j]) : "v"(ad));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t lo = (uint32_t)v[j], hi = (uint32_t)(v[j] >> 32);
            acc ^= lo + 3u * hi;
            if (check) {
                const uint32_t i0 = ((a[j] - (uint32_t)(uintptr_t)lds) >> 2) + 2u * (uint32_t)(it & 7);
                if (lo != 0x9E3779B9u * i0 + 12345u || hi != 0x9E3779B9u * (i0 + 1) + 12345u)
                    atomicAdd(&out[0], 1u);
            }
        }
    }
    out[1 + blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const int blocks = 256 * 6;  // six workgroups per CU, as the view kernel
    uint32_t *d_out, *d_addr;
    hipMalloc(&d_out, (1 + blocks * 256) * 4);
    hipMalloc(&d_addr, 8 * 256 * 4);
    struct Pat { const char* name; double step; int odd; };
    // lane l of a wave reads dwords base + floor(l * step) (+1): step = source pixels per output pixel along a row
    const Pat pats[] = {{"aligned, 2 dwords per lane (no overlap)", 2.0, 0}, {"misaligned (+1), 2 dwords per lane", 2.0, 1},
                        {"step 1.06 (config 2, view centre)", 1.06, 1}, {"step 1.36", 1.36, 1}, {"step 0.7 (view edge)", 0.7, 1}};
    for (const Pat& p : pats) {
        std::vector<uint32_t> ad(8 * 256);
        for (int j = 0; j < 8; ++j)
            for (int t = 0; t < 256; ++t) {
                const int w = t >> 6, l = t & 63;
                uint32_t dw = (uint32_t)(j * 700 + w * 150 + (int)(l * p.step));
                if (p.step == 2.0) dw = (dw & ~1u) + (p.odd ? 1u : 0u);
                ad[j * 256 + t] = dw;  // < 8192 - 16
            }
        hipMemcpy(d_addr, ad.data(), ad.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            hipMemset(d_out, 0, 4);
            if (mode == 0) k<0><<<blocks, 256>>>(d_out, d_addr, 64, 1); else k<1><<<blocks, 256>>>(d_out, d_addr, 64, 1);
            hipError_t e = hipDeviceSynchronize();
            uint32_t bad = 0;
            hipMemcpy(&bad, d_out, 4, hipMemcpyDeviceToHost);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 4000;
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(d_out, d_addr, iters, 0); else k<1><<<blocks, 256>>>(d_out, d_addr, iters, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // per CU: 6 workgroups x 4 waves x iters x 8 wave-instructions
            const double ns = ms * 1e6 / (6.0 * 4 * iters * 8);
            printf("%-44s %-14s: %s, mismatches %u, %.2f ns per wave-instruction per CU (= %.1f cycles at 2.4 GHz)\n", p.name,
                   mode ? "ds_read_b64" : "ds_read2_b32", hipGetErrorString(e), bad, ns, ns * 2.4);
        }
    }
    return 0;
}
