// p2p_audit.h -- range-checked addressing for the view kernels, and the audit build's violation record.
//
// Shipped build: every address a view kernel forms from DATA it read from memory (plan tables, yaw tables, pair
// lists) is either clamped into its buffer by a compare the kernel needs anyway, or goes through a buffer
// descriptor whose num_records is the exact extent -- the hardware drops out-of-range lanes (loads return 0,
// stores write nothing).  Indices that follow from the workgroup's own position are in range by construction.
// So whatever the tables hold, these kernels cannot leave their buffers.
//
// Audit build (-DP2P_AUDIT, tools/build_audit.sh): every such compare also RECORDS the first violation -- site,
// workgroup, thread, the offending value and its limit -- in a device word block the host reads back after
// every launch (p2p_job_run then fails with the record in p2p_last_error()).  The host side of that build also
// fills every plan pool with 0xFF before the plan pass, so that a read of a slot nobody wrote shows.
#ifndef P2P_AUDIT_H
#define P2P_AUDIT_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "p2p_device.h"

namespace p2p {
namespace P2P_SHAPE_NS {

// sites (kernel << 8 | what)
enum AuditSite : uint32_t {
    AUD_MAIN_PITCH = 0x101, AUD_MAIN_HDR = 0x102, AUD_MAIN_SRC = 0x103, AUD_MAIN_TAP = 0x104, AUD_MAIN_ITEM = 0x105,
    AUD_REST_PITCH = 0x201, AUD_REST_PAIR = 0x202, AUD_REST_SRC = 0x203, AUD_REST_F4 = 0x204, AUD_REST_TAP = 0x205,
    AUD_REST_HDR = 0x206,
    AUD_TABLE_LIST = 0x301, AUD_TABLE_PAIR = 0x302, AUD_TABLE_SRC = 0x303, AUD_TABLE_YTAB = 0x304,
    AUD_GATHER_LIST = 0x401, AUD_GATHER_BOX = 0x402, AUD_GATHER_SRC = 0x403, AUD_GATHER_COORD = 0x404,
    AUD_BAND_TILE = 0x601, AUD_BAND_GRP = 0x602, AUD_BAND_RECT = 0x603,
    AUD_FLOAT_HDR = 0x501, AUD_FLOAT_SRC = 0x502, AUD_FLOAT_LIST = 0x503, AUD_FLOAT_TAP = 0x504,
};

#ifdef P2P_AUDIT
__device__ __noinline__ void audit_record(uint32_t* rec, uint32_t site, uint32_t value, uint32_t limit)
{
    if (rec && atomicCAS(rec, 0u, 1u) == 0u) {
        rec[1] = site;
        rec[2] = blockIdx.x;
        rec[3] = blockIdx.y;
        rec[4] = blockIdx.z;
        rec[5] = threadIdx.x;
        rec[6] = value;
        rec[7] = limit;
        __threadfence();
    }
}
// value must be < limit
#define P2P_AUD_LT(rec, site, value, limit)                                        \
    do {                                                                           \
        if (!((uint32_t)(value) < (uint32_t)(limit)))                              \
            ::p2p::audit_record((rec), (site), (uint32_t)(value), (uint32_t)(limit)); \
    } while (0)
// byte range [off, off + bytes) must lie inside [0, extent)
#define P2P_AUD_RANGE(rec, site, off, bytes, extent)                                           \
    do {                                                                                       \
        if (!((uint64_t)(uint32_t)(off) + (uint64_t)(bytes) <= (uint64_t)(extent)))            \
            ::p2p::audit_record((rec), (site), (uint32_t)(off), (uint32_t)((extent) > 0xFFFFFFFFull ? 0xFFFFFFFFu : (extent))); \
    } while (0)
#else
#define P2P_AUD_LT(rec, site, value, limit) do { } while (0)
#define P2P_AUD_RANGE(rec, site, off, bytes, extent) do { } while (0)
#endif

// ---- buffer descriptors: raw buffers (stride 0), num_records = bytes; out-of-range lanes are dropped by the hardware ----
typedef unsigned int bu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int bu32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_buf(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

}  // namespace P2P_SHAPE_NS
}  // namespace p2p
#endif  // P2P_AUDIT_H
