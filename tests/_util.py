"""Shared helpers for the parity tests (fixtures, error metric)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Parity bar from BASELINE.json north_star: 1e-4 relative, fp32.  Applied
# per tensor, norm-relative (elementwise-relative is
# meaningless near zero).
RTOL = 1e-4


def load(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub(d, prefix, as_torch=True):
    out = {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}
    if as_torch:
        out = {k: torch.from_numpy(np.array(v)) for k, v in out.items()}
    return out


def rel_err(a, b):
    """max|a-b| / max|b| (per-tensor max-abs-normalised)."""
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    denom = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / denom)


def assert_close(a, b, tol=RTOL, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e


def summarize(t):
    """Same summary vector make_goldens.py stores for large tensors."""
    t = np.asarray(t.detach().cpu() if isinstance(t, torch.Tensor) else t, dtype=np.float64).ravel()
    n = t.size
    idx = (np.arange(16) * max(1, n // 16)) % n
    return np.concatenate([[n, t.sum(), np.abs(t).sum(), np.sqrt((t * t).sum())], t[:16 if n >= 16 else n], t[idx]])


def like(got, want):
    """``got`` in the form the fixture stores ``want`` in: a gradient with more than golden_recipes.BIG elements is
    kept as a strided sample of its (reference-ordered) elements -- the whole tensor otherwise."""
    from tests.golden_recipes import BIG, big_sample
    n = int(np.prod(got.shape))
    if n > BIG and tuple(got.shape) != tuple(np.shape(want)):
        s = big_sample(got)
        return torch.from_numpy(s) if isinstance(want, torch.Tensor) else s
    return got
