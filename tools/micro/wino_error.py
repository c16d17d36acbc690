"""fp32 error of 1-D Winograd F(2,3) and F(4,3) along x against a float64 direct convolution (3x3, 32 -> 32 channels,
activations ~ ReLU outputs, weights ~ the encoder's): the size of the rounding noise each algorithm adds to a
pre-activation -- what decides how many ReLU branches land on the other side of zero (tests/test_gpu_fullsize.py).
CPU only (numpy): python tools/micro/wino_error.py"""
import numpy as np

rs = np.random.RandomState(0)
B, C, H, W = 8, 32, 35, 37  # input 35 x 37 -> output 33 x 35 (rounded up to tiles by zero padding on the right)
x = np.maximum(rs.randn(B, C, H, W), 0).astype(np.float32)
w = (rs.randn(32, C, 3, 3) * 0.08).astype(np.float32)


def direct(x, w, dt):
    x, w = x.astype(dt), w.astype(dt)
    Ho, Wo = x.shape[2] - 2, x.shape[3] - 2
    out = np.zeros((x.shape[0], w.shape[0], Ho, Wo), dt)
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("bchw,oc->bohw", x[:, :, dy:dy + Ho, dx:dx + Wo], w[:, :, dy, dx]).astype(dt)
    return out


def wino(x, w, m):
    """F(m,3) along x in float32: transforms in fp32, products accumulated in fp32 over (dy, c)."""
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
    else:
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float32)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float32)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float32)
    a = m + 2
    Bn, Cc, Hh, Ww = x.shape
    Ho, Wo = Hh - 2, Ww - 2
    nt = (Wo + m - 1) // m
    xp = np.zeros((Bn, Cc, Hh, nt * m + 2), np.float32)
    xp[..., :Ww] = x
    U = np.einsum("pk,ocyk->ocyp", G, w).astype(np.float32)             # [o][c][dy][a]
    out = np.zeros((Bn, w.shape[0], Ho, nt * m), np.float32)
    for t in range(nt):
        d = xp[..., t * m:t * m + a]                                       # [B][C][H][a]
        V = np.einsum("pk,bchk->bchp", BT, d).astype(np.float32)           # transformed window
        M = np.zeros((Bn, w.shape[0], Ho, a), np.float32)
        for dy in range(3):
            M += np.einsum("bchp,ocp->bohp", V[:, :, dy:dy + Ho], U[:, :, dy]).astype(np.float32)
        out[..., t * m:(t + 1) * m] = np.einsum("jp,bohp->bohj", AT, M).astype(np.float32)
    return out[..., :Wo]


ref = direct(x, w, np.float64)
scale = np.abs(ref).max()
for name, got in (("direct fp32", direct(x, w, np.float32)), ("F(2,3) fp32", wino(x, w, 2)), ("F(4,3) fp32", wino(x, w, 4))):
    e = np.abs(got.astype(np.float64) - ref)
    print(f"{name}: max |err| {e.max():.3e}  rms {np.sqrt((e ** 2).mean()):.3e}  (max |out| {scale:.2f}; "
          f"max err / max out {e.max() / scale:.2e})")
