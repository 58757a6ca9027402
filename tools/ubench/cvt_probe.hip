// What v_cvt_pk_u8_f32 and v_dot2_f32_f16 do with the values the float pixel path feeds them (rounding, clamping).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float* outf, int n)
{
    int i = threadIdx.x;
    if (i < n) {
        out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
        h2 a = {(_Float16)(256.0f + 17.0f), (_Float16)(256.0f + 200.0f)};
        h2 w = {(_Float16)(1.0f - in[i] / 512.0f), (_Float16)(in[i] / 512.0f)};
        outf[i] = __builtin_amdgcn_fdot2(a, w, -256.0f, false);
        // bytes taken as float16 SUBNORMALS (bit pattern 0x00vv = v * 2^-24), weights scaled by 2^15, result * 2^9
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        const us2 raw = {17, 200};
        const float x = in[i] / 512.0f;
        h2 ws = {(_Float16)((1.0f - x) * 32768.0f), (_Float16)(x * 32768.0f)};
        outf[i + 32] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, raw), ws, 0.0f, false) * 512.0f;
    }
}
int main()
{
    const float v[] = {0.5f, 1.5f, 2.5f, 2.49f, 2.51f, 254.5f, 255.4f, 255.6f, 300.f, -3.f, 0.49f, 127.5f, 128.5f, 3.5f};
    const int n = sizeof(v) / sizeof(v[0]);
    float *d, *df; unsigned* o;
    hipMalloc(&d, sizeof(v)); hipMalloc(&o, n * 4); hipMalloc(&df, 64 * 4);
    hipMemcpy(d, v, sizeof(v), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, df, n);
    unsigned r[32]; float rf[64];
    hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rf, df, 64 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%8.3f -> %3u   dot2 %.4f   subnormal-dot2 %.4f (want %.4f)\n", v[i], r[i], rf[i], rf[i + 32], 17.0f + 183.0f * v[i] / 512.0f);
    return 0;
}
