import torch, time
x = torch.empty(1<<31, dtype=torch.uint8, device="cuda")
y = torch.empty(1<<31, dtype=torch.uint8, device="cuda")
for name, f, nbytes in (("fill (write only)", lambda: x.fill_(7), 1<<31), ("copy (read+write)", lambda: y.copy_(x), 1<<32), ("sum (read only)", lambda: x.view(torch.int32).sum(), 1<<31)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(name, "%.2f TB/s" % (10 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e12))
