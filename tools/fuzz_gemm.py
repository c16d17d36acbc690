#!/usr/bin/env python3
"""Random-shape parity sweep of curla_gemm (all operand layouts, batches, split-K, epilogues) against PyTorch
fp32 on the CPU.  Usage: tools/fuzz_gemm.py [n] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curla_amd import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for case in range(n):
        M = int(rs.choice([1, 3, 17, 50, 64, 100, 512, 1024]))
        N = int(rs.choice([1, 4, 50, 52, 64, 130, 1024, 3000]))
        K = int(rs.choice([2, 50, 52, 64, 96, 512, 1000, 4096]))
        if M * N * K > 3e8:
            K = max(2, int(3e8 // (M * N)))
        nb = int(rs.choice([1, 1, 2, 3]))
        ak, bk = bool(rs.randint(2)), bool(rs.randint(2))
        ks = int(rs.choice([1, 1, 1, 2, 5])) if K >= 64 else 1
        g = torch.Generator().manual_seed(case)
        A = torch.randn(nb, M, K, generator=g)
        Bm = torch.randn(nb, N, K, generator=g)
        ref = torch.einsum("zmk,znk->zmn", A, Bm)
        Ad = (A.transpose(1, 2).contiguous() if ak else A).cuda()
        Bd = (Bm.transpose(1, 2).contiguous() if bk else Bm).cuda()
        lda, ldb = (M if ak else K), (N if bk else K)
        bias = mask = None
        relu = 0
        if ks == 1:
            if rs.randint(2):
                bias = torch.randn(nb, N, generator=g)
                ref = ref + bias[:, None, :]
            if rs.randint(2):
                relu = 1
                ref = torch.relu(ref)
            if rs.randint(2):
                mask = torch.randn(nb, M, N, generator=g)
                ref = ref * (mask > 0)
        C = torch.full((ks, nb, M, N), float("nan"), device="cuda")
        ops.gemm(Ad, int(ak), lda, M * K, Bd, int(bk), ldb, N * K, C, N, M * N, M, N, K, nb, ksplit=ks,
                 split_stride=nb * M * N, bias=None if bias is None else bias.cuda(), sBias=N, relu=relu,
                 mask=None if mask is None else mask.cuda(), ldmask=N, sMask=M * N)
        got = C.sum(0).cpu()
        e = float((got - ref).abs().max() / max(1e-30, float(ref.abs().max())))
        worst = max(worst, e)
        flag = "" if e < 1e-4 else "   <-- FAIL"
        print(f"M={M:5d} N={N:5d} K={K:5d} nb={nb} A{'k' if ak else 'r'} B{'k' if bk else 'r'} ks={ks} "
              f"bias={bias is not None:d} relu={relu} mask={mask is not None:d}  err {e:.1e}{flag}", flush=True)
    print(f"worst {worst:.2e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
