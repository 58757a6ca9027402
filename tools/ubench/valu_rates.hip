// Micro-benchmark: sustained issue rate of the VALU instructions the remap kernels lean on (gfx950).
// Every block runs 256 threads x many waves per CU; each variant is a chain of 8 independent
// accumulators so that dependency latency is hidden.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
    uint32_t b = seed | 1u, c = seed ^ 0x55AA55AAu;
    unsigned long long mask64 = 0x5555555555555555ull * (seed & 3u);
    asm volatile("s_mov_b64 vcc, %0" :: "s"(mask64) : "vcc");
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8]; for (int i = 0; i < 8; ++i) { p[i].x = (float)a[i]; p[i].y = 1.0f; }
    f2 pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = __builtin_bit_cast(uint32_t, __builtin_fmaf(__builtin_bit_cast(float, a[i]), __builtin_bit_cast(float, b), __builtin_bit_cast(float, c)));
                if (OP == 1) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 4) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 5) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 6) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(a[i]) : "v"(b));
                if (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 9) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 11) asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(a[i]));
                if (OP == 12) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 13) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 14) asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
                if (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 16) asm volatile("v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == 48) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask64));
                if (OP == 49) asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == 50) asm volatile("v_cmp_gt_u32_e64 %2, %1, %0" : "+v"(a[i]) : "v"(b), "s"(mask64));
                if (OP == 51) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 18) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 19) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
                if (OP == 20) asm volatile("v_pk_lshrrev_b16 %0, 5, %0" : "+v"(a[i]));
                if (OP == 21) asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 22) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                if (OP == 23) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
                if (OP == 24) asm volatile("v_cvt_pk_u8_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 25) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
                if (OP == 26) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
                if (OP == 27) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 28) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 29) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if (OP == 30) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 31) asm volatile("v_and_b32 %0, 0xff00ff, %0" : "+v"(a[i]));
                if (OP == 32) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 33) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 34) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 35) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 36) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 37) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
                if (OP == 38) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 39) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 40) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 41) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
                if (OP == 42) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
                if (OP == 43) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 44) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
                if (OP == 45) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 46) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 47) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i] ^ __builtin_bit_cast(uint32_t, p[i].x) ^ __builtin_bit_cast(uint32_t, p[i].y);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, uint32_t* d_out, int waves_per_simd)
{
    const int cus = 256, iters = 400;
    dim3 grid(cus * waves_per_simd), block(256);  // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, block>>>(d_out, 12345u, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<grid, block>>>(d_out, 12345u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves x iters x REP instructions
    double insts = (double)waves_per_simd * iters * REP;
    double ns_per = ms * 1e6 / insts;
    printf("%-24s waves/SIMD %d: %.3f ns per wave-instr per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ns_per, ns_per * 2.4);
    return ns_per;
}

int main()
{
    uint32_t* d_out; hipMalloc(&d_out, 256 * 8 * 256 * 4);
    for (int w : {8}) {
        run<0>("v_fma_f32", d_out, w);
        run<1>("v_mad_u32_u24", d_out, w);
        run<2>("v_and_b32", d_out, w);
        run<3>("v_perm_b32", d_out, w);
        run<4>("v_pk_mad_u16", d_out, w);
        run<5>("v_dot2_u32_u16", d_out, w);
        run<6>("v_alignbyte_b32", d_out, w);
        run<7>("v_mul_u32_u24", d_out, w);
        run<8>("v_mul_lo_u32", d_out, w);
        run<9>("v_lshl_or_b32", d_out, w);
        run<10>("v_and_or_b32", d_out, w);
        run<11>("v_lshrrev_b32", d_out, w);
        run<12>("v_pk_mul_lo_u16", d_out, w);
        run<13>("v_dot4_u32_u8", d_out, w);
        run<14>("v_mul_u32_u24_sdwa", d_out, w);
        run<15>("v_add_u32", d_out, w);
        run<16>("v_mov_b32_dpp", d_out, w);
        run<17>("v_cndmask_b32", d_out, w);
        run<18>("v_add3_u32", d_out, w);
        run<19>("v_bfe_u32", d_out, w);
        run<20>("v_pk_lshrrev_b16", d_out, w);
        run<21>("v_mad_u32_u16", d_out, w);
        run<22>("v_pk_fma_f32", d_out, w);
        run<23>("v_cvt_f32_ubyte1", d_out, w);
        run<24>("v_cvt_pk_u8_f32", d_out, w);
        run<25>("v_floor_f32", d_out, w);
        run<26>("v_cvt_u32_f32", d_out, w);
        run<27>("v_mul_f32", d_out, w);
        run<28>("v_fmac_f32", d_out, w);
        run<29>("v_pk_mul_f32", d_out, w);
        run<30>("v_or_b32", d_out, w);
        run<31>("v_and_b32 literal", d_out, w);
        run<32>("v_or3_b32", d_out, w);
        run<33>("v_lshl_add_u32", d_out, w);
        run<34>("v_mad_i32_i24", d_out, w);
        run<35>("v_lerp_u8", d_out, w);
        run<36>("v_sad_u8", d_out, w);
        run<37>("v_cvt_f32_u32", d_out, w);
        run<38>("v_sub_u32", d_out, w);
        run<39>("v_max_u32", d_out, w);
        run<40>("v_mad_f32", d_out, w);
        run<41>("v_lshlrev_b32", d_out, w);
        run<42>("v_rndne_f32", d_out, w);
        run<43>("v_mov_b32", d_out, w);
        run<44>("v_and_b32 sgpr", d_out, w);
        run<45>("v_pk_add_u16", d_out, w);
        run<46>("v_add_f32", d_out, w);
        run<47>("v_mad_mix_f32", d_out, w);
        run<48>("v_cndmask_e64 sgpr mask", d_out, w);
        run<49>("v_cmp+v_cndmask pair", d_out, w);
        run<50>("v_cmp_e64 (-> sgpr)", d_out, w);
        run<51>("v_min_u32", d_out, w);
    }
    return 0;
}
