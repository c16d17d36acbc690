#!/usr/bin/env python3
"""Rounding error of one conv pre-activation (a K = 288 dot product) under the arithmetic of each stride-1 kernel form,
against float64 -- CPU model, no GPU needed:

  fp32 chain   one fmaf chain in float32: the reference's own arithmetic (PyTorch's CPU / cuDNN direct kernels differ
               in order only) and what the f32-input MFMA of conv_rw.h computes per Winograd position
  bf16x3       conv_rwb.h: every operand split into three round-to-nearest bf16 parts, the six products of weight
               >= 2^-16, each an exact bf16 x bf16 product accumulated in float32 per k = 32 slice, small terms first
  bf16x2 (3)   what dropping the third part would cost (NOT used: shown because it is the obvious cheaper variant)

python tools/micro/bf16x3_error.py
"""
import numpy as np
import torch


def split3(x):
    h = x.to(torch.bfloat16)
    r = x - h.float()
    m = r.to(torch.bfloat16)
    r2 = r - m.float()
    return h.float(), m.float(), r2.to(torch.bfloat16).float()


def mfma_chain(terms, n, K):
    """fp32 accumulator; each (a, b) term of each k = 32 slice is one matrix instruction: its 32 products are exact and
    summed (modelled in float64), the result is added to the accumulator in float32."""
    out = torch.zeros(n, terms[0][1].shape[1])
    for k0 in range(0, K, 32):
        s = slice(k0, k0 + 32)
        for a, b in terms:
            out = (out.double() + a[:, s].double() @ b[s].double()).float()
    return out


def main():
    torch.manual_seed(0)
    N, K = 8192, 288
    cases = {"activations relu(N(0, .7))": torch.relu(torch.randn(N, K)) * 0.7,
             "gradients N(0, 1) * lognormal * 1e-6": torch.randn(N, K) * torch.exp(torch.randn(N, 1) * 2) * 1e-6,
             "activations * 1e-4": torch.relu(torch.randn(N, K)) * 0.7e-4}
    w = torch.randn(K, 32) * (1.5 / np.sqrt(K))
    wh, wm, wl = split3(w)
    for name, x in cases.items():
        ref = x.double() @ w.double()
        err = lambda y: (float((y.double() - ref).abs().max() / ref.abs().max()),  # noqa: E731
                         float(((y.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()))
        chain = torch.zeros(N, 32)
        for k in range(K):
            chain = torch.addcmul(chain, x[:, k:k + 1], w[k:k + 1, :])
        xh, xm, xl = split3(x)
        six = mfma_chain([(xh, wl), (xl, wh), (xm, wm), (xh, wm), (xm, wh), (xh, wh)], N, K)
        three = mfma_chain([(xh, wm), (xm, wh), (xh, wh)], N, K)
        assert torch.equal(xh + xm + xl, x) and torch.equal(wh + wm + wl, w)  # the split is exact
        print(f"{name}\n   (max, rms) error / scale:  fp32 chain {err(chain)[0]:.2e} {err(chain)[1]:.2e}   "
              f"bf16x3 six terms {err(six)[0]:.2e} {err(six)[1]:.2e}   bf16x2 three terms {err(three)[0]:.2e} {err(three)[1]:.2e}")


if __name__ == "__main__":
    main()
