#!/usr/bin/env python3
"""Random-shape parity sweep of the first-layer kernels reading the uint8 replay ring (gather + crop + /255 +
3x3 stride-2 conv, and its weight gradient) against PyTorch fp32 on the CPU.  Usage: tools/fuzz_conv1.py [n] [seed]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curla_amd import ops  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for case in range(n):
        C = int(rs.choice([3, 6, 9, 12]))
        Hs, Ws = int(rs.randint(7, 120)), int(rs.randint(7, 150))
        Hc, Wc = int(rs.randint(5, Hs + 1)), int(rs.randint(5, Ws + 1))
        B = int(rs.choice([1, 2, 5, 9, 33]))
        N = 7
        frames = rs.randint(0, 256, (N, Hs, Ws, C), dtype=np.uint8)
        store = torch.zeros(N * Hs * Ws * C + 32, dtype=torch.uint8, device="cuda")
        store[:N * Hs * Ws * C] = torch.from_numpy(frames.reshape(-1)).cuda()
        ring = store[:N * Hs * Ws * C].view(N, Hs, Ws, C)
        idx = rs.randint(0, N, B)
        h1 = rs.randint(0, Hs - Hc + 1, B).astype(np.int32)
        w1 = rs.randint(0, Ws - Wc + 1, B).astype(np.int32)
        g = torch.Generator().manual_seed(case)
        w = torch.randn(32, C, 3, 3, generator=g) * 0.1
        b = torch.randn(32, generator=g) * 0.1
        x = torch.stack([torch.from_numpy(frames[idx[i], h1[i]:h1[i] + Hc, w1[i]:w1[i] + Wc].astype(np.float32))
                         for i in range(B)]).permute(0, 3, 1, 2) / 255.0
        wr = w.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True)
        pre = F.conv2d(x, wr, br, stride=2)
        y = torch.relu(pre)
        Ho, Wo = y.shape[2], y.shape[3]
        ref = ops.ObsRef.from_ring(ring, torch.from_numpy(idx.astype(np.int64)).cuda(), torch.from_numpy(h1).cuda(),
                                   torch.from_numpy(w1).cuda(), B, (Hc, Wc))
        out = torch.full((B, Ho, Wo, 32), float("nan"), device="cuda")
        ops.conv1_fwd(ref, w.cuda(), b.cuda(), out)
        e1 = rel(out.permute(0, 3, 1, 2).cpu(), y.detach())
        gy = torch.randn(pre.shape, generator=g)
        pre.backward(gy)
        dw, db = torch.empty(32, C, 3, 3, device="cuda"), torch.empty(32, device="cuda")
        ws = torch.empty(ops.wgrad_workspace_floats(C), device="cuda")
        ops.conv1_wgrad(ref, gy.permute(0, 2, 3, 1).contiguous().cuda(), dw, db, ws)
        e2, e3 = rel(dw.cpu(), wr.grad), rel(db.cpu(), br.grad)
        worst = max(worst, e1, e2, e3)
        flag = "" if max(e1, e2, e3) < 1e-4 else "   <-- FAIL"
        print(f"C={C:2d} ring {Hs:3d}x{Ws:3d} crop {Hc:3d}x{Wc:3d} B={B:3d}  fwd {e1:.1e}  dW {e2:.1e}  db {e3:.1e}{flag}", flush=True)
    print(f"worst {worst:.2e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
