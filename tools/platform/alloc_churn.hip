// Platform check, independent of the library: does freshly allocated device memory keep what a kernel (or a host
// copy) has just written into it, when several processes allocate and free on one GPU at the same time?
//   hipcc --offload-arch=gfx950 -O2 -o alloc_churn alloc_churn.hip ;  ./alloc_churn SECONDS [seed]
// Every round: hipMalloc (random size) -> producer (fill kernel, or hipMemcpy from pageable host memory, on a stream)
// -> stream sync -> consumer kernel on ANOTHER stream counts the words that differ -> hipFree.  Prints the rounds
// with a non-zero count: size, producer, number of wrong words, the first wrong offset and what was found there.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

__global__ void fill(uint32_t* p, size_t n, uint32_t salt)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (uint32_t)i * 2654435761u + salt;
}

__global__ void check(const uint32_t* p, size_t n, uint32_t salt, unsigned long long* res)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t want = (uint32_t)i * 2654435761u + salt, got = p[i];
        if (got != want) {
            atomicAdd(&res[0], 1ull);
            atomicMin(&res[1], (unsigned long long)i);
            if (got == 0u)
                atomicAdd(&res[2], 1ull);
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 30.0;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
    // mode 0: hipMalloc / hipFree every round; 1: one buffer allocated up front and reused (what a pool does);
    // 2: as 0, but the fresh buffer is first written once and synchronised ("touched") before the checked write
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    std::mt19937_64 rng(seed);
    hipStream_t s_prod, s_cons, s_copy;
    CK(hipStreamCreateWithFlags(&s_prod, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_cons, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
    unsigned long long* d_res;
    CK(hipMalloc((void**)&d_res, 3 * sizeof(unsigned long long)));
    std::vector<uint32_t> host;
    uint32_t* d_keep = nullptr;
    if (mode == 1)
        CK(hipMalloc((void**)&d_keep, ((size_t)1 << 24) * 4 + 4096));
    const auto t0 = std::chrono::steady_clock::now();
    long long rounds = 0, bad = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        const size_t n = (size_t)1 << (14 + rng() % 11);                  // 64 KB .. 64 MB
        const size_t words = n / 4 + rng() % 1024;
        const uint32_t salt = (uint32_t)rng();
        const int producer = (int)(rng() % 3);                            // 0 kernel, 1 pageable H2D copy, 2 both halves
        uint32_t* d = d_keep;
        if (mode != 1)
            CK(hipMalloc((void**)&d, words * 4));
        if (mode == 2) {
            hipLaunchKernelGGL(fill, dim3(512), dim3(256), 0, s_prod, d, words, ~salt);
            CK(hipStreamSynchronize(s_prod));
        }
        if (producer == 0) {
            hipLaunchKernelGGL(fill, dim3(512), dim3(256), 0, s_prod, d, words, salt);
            CK(hipStreamSynchronize(s_prod));
        } else {
            host.resize(words);
            for (size_t i = 0; i < words; ++i)
                host[i] = (uint32_t)i * 2654435761u + salt;
            const size_t half = producer == 2 ? words / 2 : words;
            CK(hipMemcpyAsync(d, host.data(), half * 4, hipMemcpyHostToDevice, s_copy));
            if (half < words)
                hipLaunchKernelGGL(fill, dim3(512), dim3(256), 0, s_prod, d, words, salt);  // (rewrites the first half too)
            CK(hipStreamSynchronize(s_copy));
            CK(hipStreamSynchronize(s_prod));
        }
        unsigned long long res[3] = {0ull, ~0ull, 0ull};
        CK(hipMemcpy(d_res, res, sizeof(res), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check, dim3(512), dim3(256), 0, s_cons, d, words, salt, d_res);
        CK(hipStreamSynchronize(s_cons));
        CK(hipMemcpy(res, d_res, sizeof(res), hipMemcpyDeviceToHost));
        if (res[0]) {
            ++bad;
            printf("WRONG round %lld: %zu words, producer %d: %llu words differ (%llu of them read 0), first at word %llu\n", rounds, words,
                   producer, res[0], res[2], res[1]);
            fflush(stdout);
        }
        if (mode != 1)
            CK(hipFree(d));
        ++rounds;
    }
    printf("alloc_churn seed %u mode %d: %lld rounds, %lld with wrong words, %.0f s\n", seed, mode, rounds, bad, seconds);
    return bad ? 1 : 0;
}
