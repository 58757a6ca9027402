#!/usr/bin/env python3
"""What plain streaming kernels reach on this GPU as a function of the buffer size (torch fill_ / copy_):
Infinity-Cache-resident sizes vs HBM, and whether very large buffers (TLB reach) cost anything.  Diagnostic."""
import sys
import torch
dev = "cuda:0"


def rate(nbytes, reps, mode):
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev) if mode == "copy" else None
    a.fill_(3)
    for _ in range(3):
        b.copy_(a) if mode == "copy" else a.fill_(1)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a) if mode == "copy" else a.fill_(1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return nbytes * (2 if mode == "copy" else 1) / ms / 1e6


sizes = [int(x) for x in sys.argv[1:]] or [8, 32, 128, 512, 2048, 4096, 8192, 16384, 24576]
for mb in sizes:
    n = mb << 20
    reps = max(3, min(200, (64 << 30) // n))
    print("%6d MB  copy %7.0f GB/s (read + write)   fill %7.0f GB/s" % (mb, rate(n, reps, "copy"), rate(n, reps, "fill")), flush=True)
