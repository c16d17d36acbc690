#!/usr/bin/env python3
"""What HBM gives a kernel that reads AND writes (the conv kernels move ~55 % reads / 45 % writes): torch's own
streaming kernels on 1 GiB tensors -- a pure read (sum), a pure write (fill), a copy (1 read : 1 write) and an add of two
tensors into a third (2 : 1) -- as GB/s of the bytes they move.  python tools/hbm_mixed.py"""
import torch

dev = "cuda"
n = 1 << 28  # floats: 1 GiB
x, y, z = (torch.empty(n, device=dev) for _ in range(3))
x.normal_(), y.normal_()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


gb = 4.0 * n / 1e9
for name, fn, moved in (("read  (sum)", lambda: x.sum(), gb), ("write (fill)", lambda: z.fill_(1.0), gb),
                        ("copy  (1 read : 1 write)", lambda: z.copy_(x), 2 * gb),
                        ("add   (2 reads : 1 write)", lambda: torch.add(x, y, out=z), 3 * gb)):
    t = timeit(fn)
    print(f"{name:28s} {t * 1e6:8.1f} us  {moved / t:8.1f} GB/s")
