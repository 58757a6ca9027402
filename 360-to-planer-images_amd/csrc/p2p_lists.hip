// p2p_lists.hip -- the main kernel's per-XCD work lists, built on the device from the plan's tile headers.
//
// The tiles the main kernel draws (mode 1), dealt to the 8 XCDs (workgroup b runs on XCD b & 7).  In the grid's own order
// -- the tile raster of one pitch view after the other -- every view reads its band of the panorama through the XCDs' L2s
// by itself, and neighbouring pitch views overlap by half (config 2: 60 / 90 / 120 degrees, 59 degrees high each): 244 MB
// of reads per launch for a 100 MB panorama and 35 MB of tables.  Here the tiles of ALL pitch views are ordered by the
// band of 64 source rows their footprint is centred in, then by view and raster position, and every XCD takes a
// contiguous part of that order: the tiles of two views that read the same rows follow each other on one XCD and find
// them in its L2 (117 MB; config 2 -2 ... -3.5 %, config 4 -4.5 %).
//
// Rounds 2-5 made these lists on the host: headers read back, a counting sort, an upload -- 75 us of a geometry's second
// launch with the GPU idle (158-162 us against 85 in the steady state).  The headers never need to leave HBM: one
// workgroup of 1024 threads does the same counting sort by source band (a band's tiles stay in (view, raster) order up to the
// turns the lanes of one LDS atomic take), the same
// running sums of the tiles' costs, the same eight cuts of equal WORK and the same choice of the end an XCD starts from.
// The host knows only the table's stride (`cap` entries per XCD, a quarter more than an equal share of all tiles) and
// sizes the grid by it; the entries each XCD really has are written next to the table (`count`), and the main kernel reads
// its XCD's count from there.
//
// Reference context: what is being scheduled is P:252-265's fan-out (one task per yaw, pitch views inside) -- here every
// (tile, chunk of yaws) of every pitch view, in the order that keeps the source rows they share in one L2.
#include "p2p_device.h"

namespace p2p {

namespace {

constexpr int LISTS_BLOCK = 1024;
constexpr int LISTS_WAVES = LISTS_BLOCK / 64;
constexpr int BATCH = 8;          // independent loads in flight per lane (see phase A)
constexpr int LISTS_COST_CAP = 12288;  // tiles whose costs fit the LDS next to the counters (24 KB of the workgroup's 57)
constexpr int LISTS_BANDS = 512;  // bands of 64 source rows: panoramas below 32767 rows (the C ABI's limit)

__device__ __forceinline__ uint32_t band_of_rows(uint32_t rows)
{
    const uint32_t b = (((rows & 0xFFFFu) + (rows >> 16)) >> 1) >> 6;
    return b < (uint32_t)LISTS_BANDS ? b : (uint32_t)LISTS_BANDS - 1u;
}

// inclusive running sum over the workgroup's 1024 threads: inside a wave by shuffles, the waves' totals through LDS
// (three barriers; a Hillis-Steele pass over LDS takes twenty)
template <class T>
__device__ __forceinline__ T block_scan_add(T v, T* s_wave_totals, int t)
{
    const int lane = t & 63, w = t >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T o = __shfl_up(v, d);
        if (lane >= d)
            v += o;
    }
    __syncthreads();  // (s_wave_totals may still be read from an earlier scan)
    if (lane == 63)
        s_wave_totals[w] = v;
    __syncthreads();
    T before = 0;
    for (int k = 0; k < LISTS_WAVES; ++k)
        before += k < w ? s_wave_totals[k] : (T)0;
    return v + before;
}

}  // namespace

__global__ __launch_bounds__(LISTS_BLOCK) void main_lists_kernel(MainListParams M)
{
    __shared__ uint32_t s_pos[LISTS_WAVES][LISTS_BANDS];  // per wave and band: count, then the wave's first position in the order
    __shared__ unsigned long long s_wtot64[LISTS_WAVES];
    __shared__ uint32_t s_wtot32[LISTS_WAVES];
    __shared__ uint32_t s_first[9], s_rev[8];
    // the tiles' costs in the sorted order, for the running sums and the cuts: in LDS when the plan is small enough (config
    // 2: 6 120 tiles), else in the block's scratch (a write and two reads of global memory between barriers)
    __shared__ uint16_t s_cost[LISTS_COST_CAP];
    const bool cost_in_lds = M.slots <= (uint32_t)LISTS_COST_CAP;
    auto cost_at = [&](uint32_t i) -> uint32_t { return cost_in_lds ? (uint32_t)s_cost[i] : M.cost[i]; };
    __shared__ unsigned long long s_want[8];
    __shared__ uint32_t s_n;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t slots = M.slots;
    // every wave a contiguous run of slots, whole chunks of 64
    const uint32_t seg = (((slots + LISTS_WAVES - 1u) / LISTS_WAVES) + 63u) & ~63u;
    const uint32_t s0 = min(slots, (uint32_t)w * seg), s1 = min(slots, s0 + seg);

    for (int i = t; i < LISTS_WAVES * LISTS_BANDS; i += LISTS_BLOCK)
        (&s_pos[0][0])[i] = 0u;
    __syncthreads();
    // A: how many mode-1 tiles of every band each wave's run holds
    // (this kernel is ONE workgroup and a chain of memory latencies: every loop below asks for BATCH independent loads
    // before it uses the first -- six dependent header reads per wave were half of its 30 us)
    auto load_batch = [&](uint32_t base, uint32_t (&mi)[BATCH], uint32_t (&rw)[BATCH]) {
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const uint32_t s = base + 64u * (uint32_t)k + (uint32_t)lane;
            mi[k] = s < s1 ? M.hdr[s].mode_items : 0u;
            rw[k] = s < s1 ? M.hdr[s].rows : 0u;
        }
    };
    // (a wave's first BATCH chunks stay in registers for the scatter below: config 2's 6120 tiles are read once)
    uint32_t mi0[BATCH], rw0[BATCH];
    load_batch(s0, mi0, rw0);
    for (uint32_t base = s0; base < s1; base += 64u * BATCH) {
        uint32_t mi[BATCH], rw[BATCH];
        if (base == s0) {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) { mi[k] = mi0[k]; rw[k] = rw0[k]; }
        } else {
            load_batch(base, mi, rw);
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            if ((mi[k] & 3u) == 1u)
                atomicAdd(&s_pos[w][band_of_rows(rw[k])], 1u);
    }
    __syncthreads();
    // B: a band's tiles start behind all tiles of the bands before it; inside a band the waves' runs follow each other
    uint32_t band_total = 0u;
    if (t < LISTS_BANDS) {
        for (int k = 0; k < LISTS_WAVES; ++k) {
            const uint32_t c = s_pos[k][t];
            s_pos[k][t] = band_total;
            band_total += c;
        }
    }
    const uint32_t upto_band = block_scan_add<uint32_t>(band_total, s_wtot32, t);  // (threads beyond the bands add 0)
    if (t < LISTS_BANDS) {
        const uint32_t start = upto_band - band_total;
        for (int k = 0; k < LISTS_WAVES; ++k)
            s_pos[k][t] += start;
    }
    if (t == LISTS_BLOCK - 1)
        s_n = upto_band;
    __syncthreads();
    const uint32_t n = s_n;
    // C: the scatter.  A wave walks its run in slot order and every lane takes its band's next position with ONE LDS atomic
    // per chunk of 64 slots (DS operations of one wave execute in order: a later chunk's tiles land behind an earlier
    // chunk's; inside a chunk the lanes of one band take their turns in the order the LDS serves them).  Round 6's first
    // version ranked the lanes of a chunk band by band -- a loop of ballots with an LDS round trip per distinct band:
    // 7.3 us of the kernel's 22.6 (tools/lists_kernel_phases.sh).
    for (uint32_t base = s0; base < s1; base += 64u * BATCH) {
        uint32_t mi[BATCH], rw[BATCH];
        if (base == s0) {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) { mi[k] = mi0[k]; rw[k] = rw0[k]; }
        } else {
            load_batch(base, mi, rw);
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const uint32_t s = base + 64u * (uint32_t)k + (uint32_t)lane;
            if ((mi[k] & 3u) == 1u) {
                const uint32_t pos = atomicAdd(&s_pos[w][band_of_rows(rw[k])], 1u);
                if (pos < slots) {
                    const uint32_t c = M.cost_base + (mi[k] >> 8);
                    M.order[pos] = s;
                    if (cost_in_lds)
                        s_cost[pos] = (uint16_t)min(c, 65535u);
                    else
                        M.cost[pos] = c;
                }
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // D: running sums of the costs along the order; the eight cuts: XCD x starts at the first position whose running sum
    // reaches x / 8 of the total (std::lower_bound over the prefix sums, as the host's version had it)
    const uint32_t per_t = (n + LISTS_BLOCK - 1u) / LISTS_BLOCK;
    const uint32_t a = min(n, (uint32_t)t * per_t), b = min(n, a + per_t);
    unsigned long long mine = 0ull;
    for (uint32_t i0 = a; i0 < b; i0 += BATCH) {
        uint32_t c[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            c[k] = i0 + (uint32_t)k < b ? cost_at(i0 + (uint32_t)k) : 0u;
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            mine += c[k];
    }
    const unsigned long long upto_mine = block_scan_add<unsigned long long>(mine, s_wtot64, t);
    if (t == LISTS_BLOCK - 1) {
        const unsigned long long total = upto_mine;
        for (int x = 0; x < 8; ++x) {
            s_want[x] = total * (unsigned long long)x / 8ull;
            s_first[x] = 0u;
        }
        s_first[8] = n;
    }
    __syncthreads();
    {
        unsigned long long run = upto_mine - mine;  // the sum before position a
        // (a thread's few positions hold a cut only if a target lies in (run, run + mine]: most threads skip the walk)
        bool any = false;
        for (int x = 1; x < 8; ++x)
            any = any || (run < s_want[x] && run + mine >= s_want[x]);
        if (any)
            for (uint32_t i0 = a; i0 < b; i0 += BATCH) {
                uint32_t c[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k)
                    c[k] = i0 + (uint32_t)k < b ? cost_at(i0 + (uint32_t)k) : 0u;
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const unsigned long long next = run + c[k];
                    for (int x = 1; x < 8; ++x)
                        if (run < s_want[x] && next >= s_want[x])
                            s_first[x] = i0 + (uint32_t)k + 1u;
                    run = next;
                }
            }
    }
    __syncthreads();
    if (t == 0) {
        // ascending, and no run longer than the grid provides for (cap), the XCDs behind still able to take the rest
        const uint32_t cap = M.cap;
        for (uint32_t x = 1; x < 8; ++x) {
            uint32_t v = min(s_first[x], n);
            v = max(v, s_first[x - 1]);
            v = min(v, s_first[x - 1] + cap);
            const uint32_t need = (8u - x) * cap;
            if (n > need)
                v = max(v, n - need);
            s_first[x] = v;
        }
    }
    __syncthreads();
    // An XCD draws its bands from the costlier end (towards a pole) to the cheaper one: the workgroups in flight when its
    // list runs out are then its shortest.  Costlier end: the one whose quarter of the run costs more.
    if (w < 8) {
        const uint32_t ra = s_first[w], rb = s_first[w + 1], q = (rb - ra) / 4u;
        unsigned long long head = 0ull, tail = 0ull;
        for (uint32_t i0 = (uint32_t)lane; i0 < q; i0 += 64u * (BATCH / 2)) {
            uint32_t ch[BATCH / 2], ct[BATCH / 2];
#pragma unroll
            for (int k = 0; k < BATCH / 2; ++k) {
                const uint32_t i = i0 + 64u * (uint32_t)k;
                ch[k] = i < q ? cost_at(ra + i) : 0u;
                ct[k] = i < q ? cost_at(rb - q + i) : 0u;
            }
#pragma unroll
            for (int k = 0; k < BATCH / 2; ++k) {
                head += ch[k];
                tail += ct[k];
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            head += __shfl_down(head, off);
            tail += __shfl_down(tail, off);
        }
        if (lane == 0)
            s_rev[w] = (q > 0u && tail > head) ? 1u : 0u;
    }
    __syncthreads();
    // E: the table, [8][cap]: an XCD's entries, then ~0 (no tile); and how many it has
    for (uint32_t i0 = (uint32_t)t; i0 < 8u * M.cap; i0 += LISTS_BLOCK * BATCH) {
        uint32_t v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const uint32_t i = i0 + (uint32_t)k * LISTS_BLOCK;
            const uint32_t x = min(i / M.cap, 7u), e = i - x * M.cap;
            const uint32_t ra = s_first[x], rb = s_first[x + 1];
            v[k] = (i < 8u * M.cap && e < rb - ra) ? M.order[s_rev[x] ? rb - 1u - e : ra + e] : ~0u;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            if (i0 + (uint32_t)k * LISTS_BLOCK < 8u * M.cap)
                M.table[i0 + (uint32_t)k * LISTS_BLOCK] = v[k];
    }
    if (t < 8)
        M.count[t] = s_first[t + 1] - s_first[t];
}

// A few words cleared by a kernel of the job's own stream: hipMemsetAsync goes through the runtime's fill kernel and the
// barrier around it (15 us in front of a cold image's plan pass, for one counter).
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        p[i] = 0u;
}

hipError_t launch_zero_words(uint32_t* p, uint32_t n, hipStream_t st)
{
    if (n == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(zero_words_kernel, dim3(std::min((n + 255u) / 256u, 1024u)), dim3(256), 0, st, p, n);
    return hipGetLastError();
}

hipError_t launch_main_lists(const MainListParams& M, hipStream_t st)
{
    hipLaunchKernelGGL(main_lists_kernel, dim3(1), dim3(LISTS_BLOCK), 0, st, M);
    return hipGetLastError();
}

}  // namespace p2p
