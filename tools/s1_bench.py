#!/usr/bin/env python3
"""Stride-1 forward stack at BASELINE configs[1] / configs[4] shapes under each value of option s1_fwd
(Winograd F(2,3) / F(4,3)): the launches update() issues -- critic phase (1024 + 512 samples, own weights) and actor /
CURL phase (512 + 512) -- timed with HIP events after a clock warm-up.  python tools/s1_bench.py [--c5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c5", action="store_true")
    ap.add_argument("--impls", default="f23,f43")
    ap.add_argument("--shapes", default="", help="B1+B2,B1+B2,... instead of the update's two launches (scaling with the batch)")
    args = ap.parse_args()
    from curla_amd import _lib, ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    H, L = (83, 5) if args.c5 else (37, 3)
    shapes = [(2048, 1024), (1024, 1024)] if args.c5 else [(1024, 512), (512, 512)]
    if args.shapes:
        shapes = [tuple(int(v) for v in sh.split("+")) for sh in args.shapes.split(",")]
    r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    w1 = [r(32, 32, 3, 3) * 0.1 for _ in range(L)]
    w2 = [r(32, 32, 3, 3) * 0.1 for _ in range(L)]
    b1 = [r(32) * 0.1 for _ in range(L)]
    b2 = [r(32) * 0.1 for _ in range(L)]
    for B1, B2 in shapes:
        x1, x2 = torch.relu(r(B1, H, H, 32)), torch.relu(r(B2, H, H, 32))
        mk = lambda B: [torch.empty(B, H - 2 * (i + 1), H - 2 * (i + 1), 32, device=dev) for i in range(L)]  # noqa: E731
        o1, o2 = mk(B1), mk(B2)
        flop = sum(2.0 * (B1 + B2) * (H - 2 * (i + 1)) ** 2 * 32 * 32 * 9 for i in range(L))
        outs = {}
        for impl in args.impls.split(","):
            with _lib.option("s1_fwd", impl):
                for _ in range(300 if not args.c5 else 30):  # clock warm-up
                    assert ops.conv_s1_fwd_stack(x1, w1, b1, o1, x2, w2, b2, o2)
                torch.cuda.synchronize()
                n = 50 if not args.c5 else 10
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    ops.conv_s1_fwd_stack(x1, w1, b1, o1, x2, w2, b2, o2)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                outs[impl] = [t.clone() for t in o1 + o2]
                print(f"s1_fwd={impl}  {B1}+{B2} x {L} layers from {H}x{H}: {us:8.1f} us  "
                      f"{flop / us / 1e6:6.1f} TF direct-equiv  {flop / us * 1e6 / PEAK * 100:5.1f} % of the f32 peak", flush=True)
        ks = list(outs)
        for k in ks[1:]:
            worst = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs[ks[0]], outs[k]))
            print(f"   max |{ks[0]} - {k}| / max |.| over all layers: {worst:.2e}")


if __name__ == "__main__":
    main()
