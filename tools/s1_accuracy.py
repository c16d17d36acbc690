#!/usr/bin/env python3
"""Error of the stride-1 forward under each value of option s1_fwd against a float64 convolution (PyTorch on the
device): max, rms and MEAN SIGNED error of the pre-ReLU... (outputs where the reference is positive), relative to the
rms of the output.  A mean far from zero relative to the rms error is a rounding bias.  python tools/s1_accuracy.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from curla_amd import _lib, ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
B, H = 64, 37
for name, x in (("relu(N(0,1)) inputs", torch.relu(torch.randn(B, 32, H, H, device=dev, generator=g))),
                ("positive inputs U(0,1), positive weights", torch.rand(B, 32, H, H, device=dev, generator=g))):
    w = torch.randn(32, 32, 3, 3, device=dev, generator=g) * 0.06
    if "positive weights" in name:
        w = w.abs()
    b = torch.randn(32, device=dev, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double())
    keep = ref > 0
    scale = float(ref[keep].pow(2).mean().sqrt())
    print(name)
    for impl in ("f23", "f43", "b3"):
        with _lib.option("s1_fwd", impl):
            out = torch.empty(B, H - 2, H - 2, 32, device=dev)
            ops.conv_s1_fwd(x.permute(0, 2, 3, 1).contiguous(), w, b, out)
        e = (out.permute(0, 3, 1, 2).double() - ref)[keep]
        print(f"   {impl}: max {float(e.abs().max()) / scale:.2e}  rms {float(e.pow(2).mean().sqrt()) / scale:.2e}  "
              f"mean {float(e.mean()) / scale:+.2e}")
    e = (F.conv2d(x, w, b).double() - ref)[keep]
    print(f"   torch fp32 conv2d: max {float(e.abs().max()) / scale:.2e}  rms {float(e.pow(2).mean().sqrt()) / scale:.2e}  "
          f"mean {float(e.mean()) / scale:+.2e}")
