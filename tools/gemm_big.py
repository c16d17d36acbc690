#!/usr/bin/env python3
"""The heads' hidden-layer products (batch x 1024 x 1024) under each arithmetic / tile of the tiled GEMM: time per
launch and the error against float64.  Usage: tools/gemm_big.py [batch=512]

  f32     gemm_mfma = f32:  the f32-input MFMA, tiles 64x64 / 64x32 / 32x32 (rounds 1-4)
  auto    the default: bf16x3 on 128 x 64 tiles where those fill the chip, f32 elsewhere
  big     gemm_tile = 12864: the 128 x 64 bf16x3 tile wherever it applies
  b3      gemm_mfma = b3:   bf16x3 with the f32 form's tiles
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curla_amd import _lib, ops  # noqa: E402

MODES = (("f32", {"gemm_mfma": "f32"}), ("auto", {}), ("big", {"gemm_tile": "12864"}), ("b3", {"gemm_mfma": "b3"}))


def timeit(fn, iters=100, warm=10):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    H = 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    R = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    print(f"# batch {B}, hidden {H}; us per launch (max relative error against float64)")
    print(f"{'product':34s}" + "".join(f"{m:>22s}" for m, _ in MODES))
    for nb in (4, 2, 1):
        x, W, b = torch.relu(R(nb, B, H)), R(nb, H, H) * 0.03, R(nb, H)
        dy = R(nb, B, H)
        out, dx, dW = torch.empty(nb, B, H, device="cuda"), torch.empty(nb, B, H, device="cuda"), torch.empty(nb, H, H, device="cuda")
        x64, W64, dy64 = x.double(), W.double(), dy.double()
        ref_fwd = torch.relu(torch.einsum("zmk,znk->zmn", x64, W64) + b.double()[:, None, :])
        ref_dx = torch.einsum("zmn,znk->zmk", dy64, W64) * (x64 > 0)
        ref_dW = torch.einsum("zmn,zmk->znk", dy64, x64)
        if nb == 4:  # the critic and the target critic, two Q functions each: a two-level batch
            # (ops.linear_fwd: the outer level's bias stride is the weights' outer stride -- biases live behind their
            # weights in the agent's flat parameter buffers)
            bflat = torch.zeros(2 * 2 * H * H, device="cuda")
            for o in range(2):
                for z in range(2):
                    bflat[o * 2 * H * H + z * H:o * 2 * H * H + (z + 1) * H] = b[2 * o + z]
            fwd = lambda: ops.linear_fwd(x, B * H, W, H * H, bflat, H, out, B * H, B, H, H, 2, relu=1,  # noqa: E731
                                         outer=(2, 2 * B * H, 2 * H * H, 2 * B * H))
        else:
            fwd = lambda: ops.linear_fwd(x, B * H, W, H * H, b, H, out, B * H, B, H, H, nb, relu=1)  # noqa: E731
        cases = [(f"fwd  NT {B}x{H}x{H} nb{nb}", fwd, lambda: [(out, ref_fwd)])]
        if nb <= 2:
            cases += [
                (f"dx   NN {B}x{H}x{H} nb{nb}",
                 lambda: ops.linear_dx(dy, B * H, W, H * H, dx, B * H, B, H, H, nb, mask=x, smask=B * H), lambda: [(dx, ref_dx)]),
                (f"dW   TN {H}x{H}x{B} nb{nb}", lambda: ops.linear_dw(dy, B * H, x, B * H, dW, H * H, B, H, H, nb),
                 lambda: [(dW, ref_dW)]),
                (f"dW + dx in one launch  nb{nb}",
                 lambda: ops.linear_bwd(dy, B * H, x, B * H, W, H * H, dW, H * H, dx, B * H, B, H, H, nb, mask=x, smask=B * H),
                 lambda: [(dW, ref_dW), (dx, ref_dx)]),
            ]
        for name, fn, refs in cases:
            row = f"{name:34s}"
            for _, opts in MODES:
                for k in ("gemm_mfma", "gemm_tile"):
                    _lib.set_option(k, opts.get(k, "auto"))
                for t, _ in refs():
                    t.zero_()
                fn()
                torch.cuda.synchronize()
                err = max(float((t.double() - r).abs().max() / r.abs().max()) for t, r in refs())
                row += f"{timeit(fn):12.1f} ({err:7.1e})"
            print(row, flush=True)
    for k in ("gemm_mfma", "gemm_tile"):
        _lib.set_option(k, "auto")


if __name__ == "__main__":
    main()
