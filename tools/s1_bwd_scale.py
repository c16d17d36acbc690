#!/usr/bin/env python3
"""One stride-1 backward launch (weight-gradient slabs + data gradient, 35 x 35 gradients) at batch sizes 128 .. 2048: its
fixed cost per launch (round 6: none at B >= 512, so merging the layers' launches buys nothing).  python tools/s1_bwd_scale.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from curla_amd import _lib, ops
def timeit(fn, iters=100, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g_ = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g_)
ws = torch.zeros(ops.wgrad_workspace_floats(32), device="cuda")
H = 35
for B in (128, 256, 512, 1024, 2048):
    x, gy, w = torch.relu(r(B, H, H, 32)), r(B, H - 2, H - 2, 32), r(32, 32, 3, 3) * 0.1
    gin = torch.empty_like(x)
    print(B, round(timeit(lambda: ops.conv_s1_bwd_slabs(x, gy, w, gin, ws)), 1), "us", flush=True)
