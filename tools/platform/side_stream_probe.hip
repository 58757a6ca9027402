// Platform check, independent of the library: when do kernels of TWO streams of one process run at the same time on this
// GPU and runtime, and what does an event between the streams do to that?
//   hipcc --offload-arch=gfx950 -O2 -o side_stream_probe side_stream_probe.hip ;  ./side_stream_probe
// Every kernel notes the device's wall clock (100 MHz) when its first workgroup starts and when its last one ends; a case
// prints each kernel's start and end in microseconds after the case's first start.
//   K0 = a 50 us kernel on the main stream (the "plan pass"), K1 = a 100 us kernel that fills the chip on the main stream
//   (the "main kernel"), K2 = a 20 us single-workgroup kernel on the side stream (the "list kernel").
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(unsigned long long* rec, unsigned long long ticks, unsigned int* done, unsigned int blocks)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0)
        atomicMin(&rec[0], t0);
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        atomicMax(&rec[1], wall_clock64());
        (void)done; (void)blocks;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv)
{
    const int k1_per_cu = argc > 1 ? atoi(argv[1]) : 6;  // workgroups of K1 per CU (8 fill every wave slot)
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    hipEvent_t e, e2;
    CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    unsigned long long* rec;
    CK(hipMalloc(&rec, 3 * 2 * sizeof(unsigned long long)));
    auto reset = [&]() {
        const unsigned long long init[6] = {~0ull, 0, ~0ull, 0, ~0ull, 0};
        return hipMemcpy(rec, init, sizeof(init), hipMemcpyHostToDevice);
    };
    auto report = [&](const char* what) {
        unsigned long long h[6];
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
            return;
        unsigned long long t0 = ~0ull;
        for (int k = 0; k < 3; ++k)
            if (h[2 * k] < t0) t0 = h[2 * k];
        printf("%-78s", what);
        for (int k = 0; k < 3; ++k)
            if (h[2 * k] != ~0ull)
                printf("  K%d %6.1f..%6.1f", k, (h[2 * k] - t0) / 100.0, (h[2 * k + 1] - t0) / 100.0);
        printf("\n");
    };
    const unsigned long long T0 = 5000, T1 = 10000, T2 = 2000;  // 100 MHz ticks
    auto K0 = [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, rec + 0, T0, nullptr, 0u); };
    auto K1 = [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(256 * k1_per_cu), dim3(256), 0, s, rec + 2, T1, nullptr, 0u); };
    auto K2 = [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(1), dim3(1024), 0, s, rec + 4, T2, nullptr, 0u); };
    for (int rep = 0; rep < 3; ++rep) {
        CK(reset()); K1(s1); K2(s2);
        report("no event: K1 on s1, K2 on s2");
        CK(reset()); K2(s2); K1(s1);
        report("no event: K2 on s2, K1 on s1");
        CK(reset()); K0(s1); CK(hipEventRecord(e, s1)); CK(hipStreamWaitEvent(s2, e, 0)); K2(s2); K1(s1);
        report("K0 s1, record, s2 waits, K2 s2, K1 s1");
        CK(reset()); K0(s1); CK(hipEventRecord(e, s1)); K1(s1); CK(hipStreamWaitEvent(s2, e, 0)); K2(s2);
        report("K0 s1, record, K1 s1, s2 waits, K2 s2");
        CK(reset()); K0(s1); CK(hipEventRecord(e, s1)); CK(hipStreamWaitEvent(s2, e, 0)); K2(s2); K1(s1);
        CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s1, e2, 0));
        report("K0 s1, record, s2 waits, K2 s2, K1 s1, s1 waits for s2");
        CK(reset()); K0(s1); CK(hipEventRecord(e, s1)); K1(s1); CK(hipStreamWaitEvent(s3, e, 0)); K2(s3);
        CK(hipEventRecord(e2, s3)); CK(hipStreamWaitEvent(s1, e2, 0));
        report("K0 s1, record, K1 s1, s3 waits, K2 s3, s1 waits for s3");
        CK(reset()); K0(s1); CK(hipStreamSynchronize(s1)); K1(s1); K2(s2);
        report("K0 s1, host waits, K1 s1, K2 s2");
    }
    return 0;
}
