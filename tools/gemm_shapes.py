#!/usr/bin/env python3
"""Times every GEMM shape of one configs[1] update (B=512, hidden 1024, F=50, fc K=30752)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curla_amd import ops


def timeit(fn, iters=50, warm=5):
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


def R(*s):
    return torch.randn(*s, device="cuda")


B, H, F, KF = 512, 1024, 50, 30752
rows = []


def rec(name, flop, fn):
    us = timeit(fn)
    rows.append((name, us, flop / us / 1e6))


for nb in (2, 1):
    x, W, b, out = R(nb, B, H), R(nb, H, H), R(nb, H), R(nb, B, H)
    rec(f"fwd  h2   NT 512x1024x1024 nb{nb}", 2 * nb * B * H * H,
        lambda: ops.linear_fwd(x, B * H, W, H * H, b, H, out, B * H, B, H, H, nb, relu=1))
    rec(f"dx   h2   NN 512x1024x1024 nb{nb}", 2 * nb * B * H * H,
        lambda: ops.linear_dx(x, B * H, W, H * H, out, B * H, B, H, H, nb, mask=x, smask=B * H))
    dW = R(nb, H, H)
    rec(f"dW   h2   TN 1024x1024x512 nb{nb}", 2 * nb * B * H * H,
        lambda: ops.linear_dw(x, B * H, out, B * H, dW, H * H, B, H, H, nb))
    k1 = 52 if nb == 2 else 50
    x1, W1 = R(nb, B, k1), R(nb, H, k1)
    rec(f"fwd  h1   NT 512x1024x{k1} nb{nb}", 2 * nb * B * H * k1,
        lambda: ops.linear_fwd(x1, B * k1, W1, H * k1, b, H, out, B * H, B, H, k1, nb, relu=1))
    dW1 = R(nb, H, k1)
    rec(f"dW   h1   TN 1024x{k1}x512 nb{nb}", 2 * nb * B * H * k1,
        lambda: ops.linear_dw(out, B * H, x1, B * k1, dW1, H * k1, B, H, k1, nb))
    dx1 = R(nb, B, k1)
    rec(f"dx   h1   NN 512x{k1}x1024 nb{nb}", 2 * nb * B * H * k1,
        lambda: ops.linear_dx(out, B * H, W1, H * k1, dx1, B * k1, B, H, k1, nb))
    n3 = 1 if nb == 2 else 4  # last layer: Q -> 1, actor trunk -> 2|A| = 4
    W3, b3, o3 = R(nb, n3, H), R(nb, n3), R(nb, B, n3)
    rec(f"fwd  out  NT 512x{n3}x1024 nb{nb}", 2 * nb * B * H * n3,
        lambda: ops.linear_fwd(x, B * H, W3, n3 * H, b3, n3, o3, B * n3, B, n3, H, nb))
    dW3 = R(nb, n3, H)
    rec(f"dW   out  TN {n3}x1024x512 nb{nb}", 2 * nb * B * H * n3,
        lambda: ops.linear_dw(o3, B * n3, x, B * H, dW3, n3 * H, B, n3, H, nb))
    rec(f"dx   out  NN 512x1024x{n3} nb{nb}", 2 * nb * B * H * n3,
        lambda: ops.linear_dx(o3, B * n3, W3, n3 * H, out, B * H, B, n3, H, nb, mask=x, smask=B * H))
h, Wfc, part = R(B, KF), R(F, KF), R(64, B, F)
for ks in (32, 64):
    rec(f"fc   fwd  NT 512x50x30752 ks{ks}", 2 * B * F * KF,
        lambda: ops.gemm(h, 0, KF, 0, Wfc, 0, KF, 0, part, F, 0, B, F, KF, 1, ksplit=ks, split_stride=B * F))
dfc, g = R(B, F), R(B, KF)
rec("fc   dx   NN 512x30752x50", 2 * B * F * KF, lambda: ops.linear_dx(dfc, 0, Wfc, 0, g, 0, B, F, KF, mask=h))
dWfc = R(F, KF)
rec("fc   dW   TN 50x30752x512", 2 * B * F * KF, lambda: ops.linear_dw(dfc, 0, h, 0, dWfc, 0, B, F, KF))
za, zp, Wc, lg = R(B, F), R(B, F), R(F, F), R(B, B)
rec("curl logits NT 512x512x50", 2 * B * B * F, lambda: ops.gemm(za, 0, F, 0, zp, 0, F, 0, lg, B, 0, B, B, F, 1))
rec("curl WzT    NT 512x50x50", 2 * B * F * F, lambda: ops.linear_fwd(zp, 0, Wc, 0, None, 0, za, 0, B, F, F))
rec("curl dz_a   NN 512x50x512", 2 * B * B * F, lambda: ops.linear_dx(lg, 0, zp, 0, za, 0, B, B, F))
rec("curl dWzT   TN 512x50x512", 2 * B * B * F, lambda: ops.linear_dw(lg, 0, za, 0, zp, 0, B, B, F))
dWc = R(F, F)
rec("curl dW     TN 50x50x512", 2 * B * F * F, lambda: ops.linear_dw(za, 0, zp, 0, dWc, 0, B, F, F))
tot = 0
for name, us, tf in rows:
    print(f"{name:36s} {us:8.1f} us {tf:7.1f} TF")
    tot += us
print(f"sum {tot:.1f} us")
