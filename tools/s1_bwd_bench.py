#!/usr/bin/env python3
"""Stride-1 backward of one layer (weight-gradient slabs + data gradient, one launch) and the weight gradient alone, per
option s1_wgrad: tools/s1_bwd_bench.py [--c5]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curla_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=100, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    c5 = "--c5" in sys.argv
    B, Hs = (1024, (81, 79, 77, 75, 73)) if c5 else (512, (37, 35, 33))
    g_ = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g_)  # noqa: E731
    ws = torch.zeros(ops.wgrad_workspace_floats(32), device="cuda")
    for H in Hs:
        x, gy, w = torch.relu(r(B, H, H, 32)), r(B, H - 2, H - 2, 32), r(32, 32, 3, 3) * 0.1
        gin = torch.empty_like(x)
        ref = None
        row = f"B={B} {H}x{H}:"
        for mode in ("x", "xy"):
            with _lib.option("s1_wgrad", mode):
                t_w = timeit(lambda: ops.conv_s1_wgrad_slabs(x, gy, ws))
                t_b = timeit(lambda: ops.conv_s1_bwd_slabs(x, gy, w, gin, ws))
                dw, db = torch.empty(32, 32, 3, 3, device="cuda"), torch.empty(32, device="cuda")
                n = ops.conv_s1_bwd_slabs(x, gy, w, gin, ws)
                ops.wgrad_reduce_multi([(ws, n, dw, db)])
                if ref is None:
                    ref = dw.clone()
                err = float((dw - ref).abs().max() / ref.abs().max())
            row += f"   s1_wgrad={mode}: wgrad alone {t_w:7.1f} us, wgrad + dgrad {t_b:7.1f} us (dW vs x: {err:.1e})"
        print(row, flush=True)


if __name__ == "__main__":
    main()
