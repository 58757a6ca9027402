#!/usr/bin/env python3
"""Platform check, independent of the library: what do hipMalloc and hipFree cost, by size, on this GPU and runtime?
    python3 tools/platform/malloc_probe.py            (GPU box; straight through libamdhip64.so with ctypes)
MI355X / ROCm 7.2.0: hipMalloc 1 us below 8 MB (the runtime's own sub-allocator), 10-13 us from 8 MB up WHATEVER the size
(25 us the first time), hipFree 125-135 us.  Why the library keeps its device memory in a pool and asks the driver for a
plan's block while the device is busy with the job's yaw tables (csrc/p2p_host_plan.cpp: plan_block_prefetch)."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes=[ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]; hip.hipFree.argtypes=[ctypes.c_void_p]
hip.hipDeviceSynchronize()
p=ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1<<20); hip.hipFree(p)
for mb in (0.004, 1, 8, 42, 92, 224, 42, 92):
    ts=[]
    for _ in range(5):
        t=time.perf_counter(); hip.hipMalloc(ctypes.byref(p), int(mb*(1<<20))); ts.append((time.perf_counter()-t)*1e6)
        t=time.perf_counter(); hip.hipFree(p); tf=(time.perf_counter()-t)*1e6
    print("hipMalloc %8.3f MB: %s us; last hipFree %.0f us" % (mb, " ".join("%.0f" % x for x in ts), tf))
# many live allocations (no free in between)
ps=[]
for mb in (42, 92):
    t=time.perf_counter()
    for _ in range(4):
        q=ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(q), int(mb*(1<<20))); ps.append(q)
    print("4 live allocations of %d MB: %.0f us each" % (mb, (time.perf_counter()-t)*1e6/4))
