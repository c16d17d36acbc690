#!/usr/bin/env python3
"""Random-shape parity sweep of the stride-1 conv kernels (forward, data gradient, weight gradient) and the
first-layer kernels against PyTorch fp32 on the CPU.  Usage: tools/fuzz_conv.py [n_cases] [seed] [s1_fwd]
(s1_fwd = auto | f23 | f43: the Winograd form of the forward and the data gradient, option s1_fwd; widths up to 200 so
that `auto` takes both)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curla_amd import _lib, ops  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    if len(sys.argv) > 3:
        _lib.set_option("s1_fwd", sys.argv[3])
    print("s1_fwd =", _lib.get_option("s1_fwd"))
    worst = 0.0
    for case in range(n):
        B = int(rs.choice([1, 2, 3, 5, 8, 17, 64, 130]))
        H, W = int(rs.randint(3, 60)), int(rs.randint(3, 200))
        if B * H * W > 300000:
            B = max(1, 300000 // (H * W))
        g = torch.Generator().manual_seed(case)
        x = torch.randn(B, 32, H, W, generator=g)
        w = torch.randn(32, 32, 3, 3, generator=g) * 0.1
        b = torch.randn(32, generator=g) * 0.1
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = torch.relu(F.conv2d(xr, wr, br))
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        xd = x.permute(0, 2, 3, 1).contiguous().cuda()
        out = torch.full((B, H - 2, W - 2, 32), float("nan"), device="cuda")
        ops.conv_s1_fwd(xd, w.cuda(), b.cuda(), out)
        e1 = rel(out.permute(0, 3, 1, 2).cpu(), y.detach())
        gpre = (gy * (y > 0)).permute(0, 2, 3, 1).contiguous().cuda()  # gradient at the pre-activation
        below = torch.relu(torch.randn(B, H, W, 32, generator=g)).cuda()
        gin = torch.full((B, H, W, 32), float("nan"), device="cuda")
        ops.conv_s1_dgrad(gpre, w.cuda(), below, gin)
        ref_gin = xr.grad.permute(0, 2, 3, 1) * (below.cpu() > 0)
        e2 = rel(gin.cpu(), ref_gin)
        dw, db = torch.empty(32, 32, 3, 3, device="cuda"), torch.empty(32, device="cuda")
        ws = torch.empty(ops.wgrad_workspace_floats(32), device="cuda")
        ops.conv_s1_wgrad(xd, gpre, dw, db, ws)
        e3, e4 = rel(dw.cpu(), wr.grad), rel(db.cpu(), br.grad)
        worst = max(worst, e1, e2, e3, e4)
        flag = "" if max(e1, e2, e3, e4) < 1e-4 else "   <-- FAIL"
        print(f"B={B:4d} {H:3d}x{W:3d}  fwd {e1:.1e}  dgrad {e2:.1e}  dW {e3:.1e}  db {e4:.1e}{flag}", flush=True)
    print(f"worst {worst:.2e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
