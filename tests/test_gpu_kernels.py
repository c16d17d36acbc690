"""Kernel-level parity on the MI355X: every HIP entry point against a plain
PyTorch fp32 (CPU) statement of the same op, through the C ABI.  Integer work
(crop, frame store) bit-exact; floating point within 1e-4 per tensor
(tests/_util.RTOL).  Shapes include ragged tiles, rectangular images and the
BASELINE geometries (84->76, 37/35/33 feature maps, 83-wide maps of config 5)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests._util import RTOL, rel_err

pytestmark = pytest.mark.gpu

REPORT = []


def check(name, got, ref, tol=RTOL):
    e = rel_err(got, ref)
    REPORT.append((name, e))
    assert np.isfinite(e) and e <= tol, f"{name}: rel err {e:.3e} > {tol:.1e}"


@pytest.fixture(scope="module", autouse=True)
def _report():
    yield
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/kernel_parity.txt", "a") as f:
        for n, e in REPORT:
            f.write(f"{n:60s} {e:.3e}\n")


@pytest.fixture(scope="module")
def ops():
    from curla_amd import ops as o
    return o


def dev(x):
    return torch.as_tensor(x).cuda().contiguous()


def nhwc(x):  # NCHW cpu -> NHWC cuda
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(x):  # NHWC cuda -> NCHW cpu
    return x.permute(0, 3, 1, 2).contiguous().cpu()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(params=["f23", "f43", "b3"])
def s1_impl(request):
    """The stride-1 forwards / data gradients: Winograd F(2,3) (conv_rw.h), F(4,3) (conv_rw43.h) and the bf16x3 form on
    the bf16 matrix cores (conv_rwb.h), option s1_fwd."""
    from curla_amd import _lib
    with _lib.option("s1_fwd", request.param):
        yield request.param


@pytest.fixture(params=["x", "xy"])
def s1_wgrad(request):
    """The stride-1 weight gradients: Winograd F(3,2) along x (conv_rw_wgrad.h) and in both directions (conv_rw_wgrad2.h),
    option s1_wgrad."""
    from curla_amd import _lib
    with _lib.option("s1_wgrad", request.param):
        yield request.param


@pytest.mark.parametrize("B,H,W", [(3, 13, 16), (2, 37, 37), (5, 35, 35), (1, 83, 83), (2, 5, 41), (9, 17, 15), (3, 6, 7),
                                   (2, 3, 3), (1, 21, 130)])
def test_conv_s1_fwd(ops, s1_impl, B, H, W):
    x, w, b = rnd(B, 32, H, W, seed=1), rnd(32, 32, 3, 3, seed=2, scale=0.1), rnd(32, seed=3, scale=0.1)
    ref = torch.relu(F.conv2d(x, w, b))
    out = torch.full((B, H - 2, W - 2, 32), float("nan"), device="cuda")
    ops.conv_s1_fwd(nhwc(x), dev(w), dev(b), out)
    check(f"conv_s1_fwd [{s1_impl}] B{B} {H}x{W}", nchw(out), ref)


@pytest.mark.parametrize("B,H,W", [(3, 11, 14), (2, 35, 35), (1, 81, 81), (4, 3, 39), (7, 13, 13)])
def test_conv_s1_dgrad(ops, s1_impl, B, H, W):
    # g: gradient w.r.t. the conv output [B,32,H,W]; input was [B,32,H+2,W+2]
    below = rnd(B, 32, H + 2, W + 2, seed=4)
    act_below = torch.relu(below)
    w = rnd(32, 32, 3, 3, seed=5, scale=0.1)
    g = rnd(B, 32, H, W, seed=6)
    ref = F.conv_transpose2d(g, w) * (act_below > 0)
    gin = torch.full((B, H + 2, W + 2, 32), float("nan"), device="cuda")
    ops.conv_s1_dgrad(nhwc(g), dev(w), nhwc(act_below), gin)
    check(f"conv_s1_dgrad B{B} {H}x{W}", nchw(gin), ref)


# (rows of >= 8 pixel pairs take the scalar pair walk, even and odd widths differently: 18, 26, 102 / 17, 19, 37 ...)
@pytest.mark.parametrize("B,H,W", [(3, 13, 16), (2, 37, 37), (6, 35, 35), (1, 83, 83), (300, 9, 9), (3, 12, 18),
                                   (2, 20, 26), (2, 19, 19), (1, 40, 102), (2, 5, 17), (4, 3, 40), (5, 4, 33), (600, 37, 37),
                                   (2, 4, 4), (1, 3, 3)])
def test_conv_s1_wgrad(ops, s1_wgrad, B, H, W):
    x = torch.relu(rnd(B, 32, H, W, seed=7))
    g = rnd(B, 32, H - 2, W - 2, seed=8) * (rnd(B, 32, H - 2, W - 2, seed=9) > 0)
    w = torch.zeros(32, 32, 3, 3, requires_grad=True)
    b = torch.zeros(32, requires_grad=True)
    F.conv2d(x, w, b).backward(g)
    dw = torch.full((32, 32, 3, 3), float("nan"), device="cuda")
    db = torch.full((32,), float("nan"), device="cuda")
    ws = torch.empty(ops.wgrad_workspace_floats(32), device="cuda")
    ops.conv_s1_wgrad(nhwc(x), nhwc(g), dw, db, ws)
    check(f"conv_s1_wgrad [{s1_wgrad}] dW B{B} {H}x{W}", dw.cpu(), w.grad)
    check(f"conv_s1_wgrad [{s1_wgrad}] db B{B} {H}x{W}", db.cpu(), b.grad)


def _ring(N, C, Hs, Ws, seed):
    frames = np.random.RandomState(seed).randint(0, 256, (N, C, Hs, Ws), dtype=np.uint8)  # CHW like the reference
    store = torch.zeros(N * C * Hs * Ws + 32, dtype=torch.uint8, device="cuda")
    ring = store[:N * C * Hs * Ws].view(N, Hs, Ws, C)
    ring.copy_(torch.from_numpy(frames).permute(0, 2, 3, 1))
    return frames, ring


def _poison_lds(ops):
    """Leave NaN bit patterns (0xFF bytes) in the LDS of every CU: the banded uint8 forward keeps its crops as bytes in
    LDS, so a ring full of 255s does it.  A kernel that reads LDS it has not written (and multiplies it by a zero
    weight) then produces NaNs deterministically instead of once in a few hundred runs."""
    from curla_amd import _lib
    with _lib.option("conv1_u8", "hybrid"):
        store = torch.full((84 * 84 * 9 + 32,), 255, dtype=torch.uint8, device="cuda")
        ring = store[:84 * 84 * 9].view(1, 84, 84, 9)
        B = 1024
        z64, z32 = torch.zeros(B, dtype=torch.int64, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda")
        obs = ops.ObsRef.from_ring(ring, z64, z32, z32, B, (84, 84))
        out = torch.empty(B, 41, 41, 32, device="cuda")
        ops.conv1_fwd(obs, torch.zeros(32, 9, 3, 3, device="cuda"), torch.zeros(32, device="cuda"), out)
        torch.cuda.synchronize()


CONV1_CASES = [  # C, Hs, Ws, Hc, Wc, B
    (9, 34, 40, 28, 34, 8), (9, 84, 84, 76, 76, 4), (9, 84, 84, 84, 84, 3), (12, 50, 46, 41, 37, 5), (3, 20, 23, 17, 19, 6),
    (6, 31, 45, 25, 39, 7),  # frame_stack 2; 31*45*6 bytes per frame is not a multiple of 4
    (3, 21, 21, 21, 21, 5),  # odd-sized frames, no crop
    (9, 76, 135, 76, 135, 4),  # the reference's own thesis shape (encoder.py:42-43), no crop; row pitch 1215 = 3 mod 4
    (12, 37, 41, 30, 33, 9),  # row pitch 492 = 0 mod 4, odd crop origins: aligned loads + byte shift
]


@pytest.fixture(params=["hybrid", "band", "rw", "rwb"])
def u8_impl(request):
    """The three uint8 first-layer forwards: the hybrid (crop staged in LDS + row walk out of LDS when the crop fits
    one band, else the banded loop), the banded loop alone (option conv1_u8 = band) and the row walk straight from
    memory, on the f32-input MFMA (conv1_u8 = rw) and on the bf16 matrix cores (conv1_u8 = rwb: what `auto` takes where
    3 C <= 32; conv1_u8_rw.h)."""
    from curla_amd import _lib
    with _lib.option("conv1_u8", request.param):
        yield request.param


@pytest.mark.parametrize("C,Hs,Ws,Hc,Wc,B", CONV1_CASES)
def test_crop_and_conv1_u8(ops, u8_impl, C, Hs, Ws, Hc, Wc, B):
    from oracle import curla_oracle as O
    N = 11
    frames, ring = _ring(N, C, Hs, Ws, seed=C + Hs)
    rs = np.random.RandomState(5)
    idx = rs.randint(0, N, B)
    h1 = rs.randint(0, Hs - Hc + 1, B).astype(np.int32)
    w1 = rs.randint(0, Ws - Wc + 1, B).astype(np.int32)
    ref_crop = O.random_crop(frames[idx], h1, w1, (Hc, Wc))
    d_idx, d_h1, d_w1 = dev(idx.astype(np.int64)), dev(h1), dev(w1)
    # integer path: bit-exact
    out_u8 = torch.zeros((B, C, Hc, Wc), dtype=torch.uint8, device="cuda")
    out_f = torch.zeros((B, C, Hc, Wc), dtype=torch.float32, device="cuda")
    ops.crop_nchw(ring, d_idx, d_h1, d_w1, B, (Hc, Wc), out_f32=out_f, out_u8=out_u8)
    assert np.array_equal(out_u8.cpu().numpy(), ref_crop)
    assert np.array_equal(out_f.cpu().numpy(), ref_crop.astype(np.float32))
    # fused gather + crop + /255 + conv1 + relu
    w, b = rnd(32, C, 3, 3, seed=11, scale=0.2), rnd(32, seed=12, scale=0.1)
    x = torch.from_numpy(ref_crop.astype(np.float32))
    ref = torch.relu(F.conv2d(x / 255.0, w, b, stride=2))
    Ho, Wo = ref.shape[2:]
    obs = ops.ObsRef.from_ring(ring, d_idx, d_h1, d_w1, B, (Hc, Wc))
    out = torch.full((B, Ho, Wo, 32), float("nan"), device="cuda")
    ops.conv1_fwd(obs, dev(w), dev(b), out)
    check(f"conv1_fwd u8 [{u8_impl}] C{C} {Hs}x{Ws}->{Hc}x{Wc}", nchw(out), ref)
    # float NCHW source (reference tensor contract); with NaN patterns in whatever LDS the kernel does not write itself
    # (at an odd crop width the last pixel's padded k-step reads one float behind the row: it must be a zero, not
    # "anything times a zero weight")
    _poison_lds(ops)
    out2 = torch.full((B, Ho, Wo, 32), float("nan"), device="cuda")
    ops.conv1_fwd(ops.ObsRef.from_tensor(out_f), dev(w), dev(b), out2)
    check(f"conv1_fwd f32 C{C} {Hc}x{Wc}", nchw(out2), ref)
    # float NHWC source (the augmented minibatches): the row walk (conv1_rw.h, the default) and the banded form
    from curla_amd import _lib
    obs_nhwc = ops.ObsRef.from_nhwc(out_f.permute(0, 2, 3, 1).contiguous())
    f32_impls = ("rw", "band") if u8_impl == "band" else ("rw",)
    for f32_impl in f32_impls:
        with _lib.option("conv1_f32", f32_impl):
            _poison_lds(ops)
            out3 = torch.full((B, Ho, Wo, 32), float("nan"), device="cuda")
            ops.conv1_fwd(obs_nhwc, dev(w), dev(b), out3)
            check(f"conv1_fwd f32 NHWC [{f32_impl}] C{C} {Hc}x{Wc}", nchw(out3), ref)
    # weight gradient, both sources
    g = rnd(B, 32, Ho, Wo, seed=13) * (ref > 0)
    wl = w.clone().requires_grad_(True)
    bl = b.clone().requires_grad_(True)
    F.conv2d(x / 255.0, wl, bl, stride=2).backward(g)
    ws = torch.empty(ops.wgrad_workspace_floats(C), device="cuda")
    # (uint8 source: on the f32-input MFMA and on the bf16 matrix cores -- option wgrad1_u8; `auto` = b16 where it applies)
    for name, o, impl, wg in [("u8 [wgrad1_u8=f32]", obs, "rw", "f32"), ("u8 [wgrad1_u8=b16]", obs, "rw", "b16"),
                              ("f32", ops.ObsRef.from_tensor(out_f), "rw", "auto")] + \
                             [(f"f32 NHWC [{i}]", obs_nhwc, i, "auto") for i in f32_impls]:
        with _lib.option("conv1_f32", impl), _lib.option("wgrad1_u8", wg):
            _poison_lds(ops)
            dw = torch.full((32, C, 3, 3), float("nan"), device="cuda")
            db = torch.full((32,), float("nan"), device="cuda")
            ops.conv1_wgrad(o, nhwc(g), dw, db, ws)
        check(f"conv1_wgrad dW {name} C{C} {Hc}x{Wc}", dw.cpu(), wl.grad)
        check(f"conv1_wgrad db {name} C{C} {Hc}x{Wc}", db.cpu(), bl.grad)


def test_store_frame_bit_exact(ops):
    C, H, W = 9, 34, 40
    ring = torch.zeros((5, H, W, C), dtype=torch.uint8, device="cuda")
    f = np.random.RandomState(2).randint(0, 256, (C, H, W), dtype=np.uint8)
    ops.store_frame(dev(f).view(-1), ring, 3)
    assert np.array_equal(ring[3].cpu().numpy(), f.transpose(1, 2, 0))
    assert int(ring[2].sum()) == 0 and int(ring[4].sum()) == 0


GEMM_CASES = [  # M, N, K, a_kmajor, b_kmajor, nbatch
    (8, 50, 2240, 0, 0, 1), (70, 64, 50, 0, 0, 1), (512, 1024, 52, 0, 0, 2), (33, 17, 129, 0, 1, 1), (50, 301, 64, 1, 1, 1),
    (64, 52, 100, 1, 1, 2), (5, 4, 64, 0, 0, 1), (128, 128, 128, 0, 1, 2), (1, 64, 50, 0, 0, 1),
]


@pytest.mark.parametrize("M,N,K,ak,bk,nb", GEMM_CASES)
def test_gemm(ops, M, N, K, ak, bk, nb):
    A = rnd(nb, M, K, seed=21)
    Bm = rnd(nb, N, K, seed=22)
    bias = rnd(nb, N, seed=23)
    mask = rnd(nb, M, N, seed=24)
    ref_plain = torch.einsum("zmk,znk->zmn", A, Bm)
    Ad = dev(A.transpose(1, 2)) if ak else dev(A)
    Bd = dev(Bm.transpose(1, 2)) if bk else dev(Bm)
    lda, ldb = (M if ak else K), (N if bk else K)
    C = torch.full((nb, M, N), float("nan"), device="cuda")
    ops.gemm(Ad, ak, lda, M * K, Bd, bk, ldb, N * K, C, N, M * N, M, N, K, nb, alpha=0.5, bias=dev(bias), sBias=N, relu=1)
    check(f"gemm bias+relu {M}x{N}x{K} ak{ak} bk{bk} nb{nb}", C.cpu(), torch.relu(0.5 * ref_plain + bias[:, None, :]))
    C2 = torch.full((nb, M, N), float("nan"), device="cuda")
    ops.gemm(Ad, ak, lda, M * K, Bd, bk, ldb, N * K, C2, N, M * N, M, N, K, nb, mask=dev(mask), ldmask=N, sMask=M * N)
    check(f"gemm mask {M}x{N}x{K} ak{ak} bk{bk} nb{nb}", C2.cpu(), ref_plain * (mask > 0))


@pytest.mark.parametrize("ak,bk", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K,nb,tile", [(128, 64, 128, 1, "12864"), (256, 192, 160, 3, "12864"), (512, 1024, 1024, 4, "auto"),
                                           (384, 128, 512, 2, "12864")])
def test_gemm_bf16x3_128x64_tile(ops, M, N, K, nb, tile, ak, bk):
    """The 128 x 64 tile of the tiled GEMM (512 threads, fp32 operands as three bf16 parts on the bf16 matrix cores; what
    `auto` takes where it gives every CU a workgroup -- the last but one case -- and gemm_tile = 12864 forces wherever whole
    tiles fit): every operand layout, batches, the bias + ReLU and the mask epilogues, against float64."""
    from curla_amd import _lib
    A, Bm = rnd(nb, M, K, seed=51), rnd(nb, N, K, seed=52)
    bias, mask = rnd(nb, N, seed=53), rnd(nb, M, N, seed=54)
    ref = torch.einsum("zmk,znk->zmn", A.double(), Bm.double())
    Ad = dev(A.transpose(1, 2)) if ak else dev(A)
    Bd = dev(Bm.transpose(1, 2)) if bk else dev(Bm)
    lda, ldb = (M if ak else K), (N if bk else K)
    with _lib.option("gemm_tile", tile):
        C = torch.full((nb, M, N), float("nan"), device="cuda")
        ops.gemm(Ad, ak, lda, M * K, Bd, bk, ldb, N * K, C, N, M * N, M, N, K, nb, alpha=0.5, bias=dev(bias), sBias=N, relu=1)
        C2 = torch.full((nb, M, N), float("nan"), device="cuda")
        ops.gemm(Ad, ak, lda, M * K, Bd, bk, ldb, N * K, C2, N, M * N, M, N, K, nb, mask=dev(mask), ldmask=N, sMask=M * N)
        with _lib.option("gemm_mfma", "f32"):  # (the option that keeps everything on the f32-input MFMA)
            C3 = torch.full((nb, M, N), float("nan"), device="cuda")
            ops.gemm(Ad, ak, lda, M * K, Bd, bk, ldb, N * K, C3, N, M * N, M, N, K, nb, mask=dev(mask), ldmask=N, sMask=M * N)
    check(f"gemm 128x64 bias+relu {M}x{N}x{K} ak{ak} bk{bk} nb{nb}", C.cpu(),
          torch.relu(0.5 * ref + bias.double()[:, None, :]).float(), 2e-5)
    check(f"gemm 128x64 mask {M}x{N}x{K} ak{ak} bk{bk} nb{nb}", C2.cpu(), (ref * (mask > 0)).float(), 2e-5)
    check(f"gemm f32 mask {M}x{N}x{K} ak{ak} bk{bk} nb{nb}", C3.cpu(), (ref * (mask > 0)).float(), 2e-5)
    assert not torch.equal(C2, C3)  # (two arithmetics: the tile was really taken)


@pytest.mark.parametrize("ak,bk", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K,nb", [(512, 54, 1024, 2), (50, 50, 512, 1), (37, 70, 256, 3), (1024, 54, 512, 2),
                                      (16, 16, 64 * 5, 1), (512, 50, 512, 1)])
def test_gemm_small_output_long_k(ops, M, N, K, ak, bk, nb):
    """Small outputs with a long k (first-layer data / weight gradients of the MLPs, the CURL bilinear products) take
    gemm_small_kernel: one workgroup per 16 x 16 tile, k split over its waves."""
    A, Bm = rnd(nb, M, K, seed=41), rnd(nb, N, K, seed=42)
    ref = torch.einsum("zmk,znk->zmn", A.double(), Bm.double()).float()
    Ad = dev(A.transpose(1, 2)) if ak else dev(A)
    Bd = dev(Bm.transpose(1, 2)) if bk else dev(Bm)
    lda, ldb = (M if ak else K), (N if bk else K)
    C = torch.full((nb, M, N + 3), float("nan"), device="cuda")  # (ldc = N + 3: rows not 16-byte aligned)
    ops.gemm(Ad, ak, lda, M * K, Bd, bk, ldb, N * K, C, N + 3, M * (N + 3), M, N, K, nb, alpha=0.25)
    check(f"small gemm {M}x{N}x{K} ak{ak} bk{bk}", C[:, :, :N].cpu(), 0.25 * ref, 2e-5)
    assert torch.isnan(C[:, :, N:]).all()
    C2 = torch.full((nb, M, N), float("nan"), device="cuda")
    ops.gemm(Ad, ak, lda, M * K, Bd, bk, ldb, N * K, C2, N, M * N, M, N, K, nb)
    check(f"small gemm {M}x{N}x{K} ak{ak} bk{bk} dense C", C2.cpu(), ref, 2e-5)
    if ak and bk:  # the weight-gradient form dy^T x with the bias gradient (row sums of opA) alongside
        assert ops.linear_dw_folds_bias(K, M, N, nb)
        C3, cs = torch.full_like(C2, float("nan")), torch.full((nb, M + 2), float("nan"), device="cuda")
        ops.linear_dw(Ad, M * K, Bd, N * K, C3, M * N, K, M, N, nb, colsum=cs, s_colsum=M + 2)
        assert torch.equal(C3, C2)
        check(f"small gemm colsum {M}x{N}x{K}", cs[:, :M].cpu(), A.double().sum(2).float(), 2e-5)
        assert torch.isnan(cs[:, M:]).all()


# B, N (out features), K (in features), nbatch, shared x, bias gradient
LINEAR_BWD_CASES = [(512, 1024, 1024, 2, False, False),  # twin-Q hidden layer: tiled pair 64x64 + 64x32 / bf16x3 128x64 (auto)
                    (512, 1024, 1024, 1, False, False),  # actor hidden layer: tiled pair 64x32 + 32x32
                    (1024, 1024, 1024, 2, False, False),  # batch 1024: 64x64 + 64x64 / bf16x3 128x64 (auto)
                    (1024, 1024, 1024, 1, False, False),  # 64x32 + 64x32 / bf16x3 128x64 (auto)
                    (512, 1024, 54, 2, True, True),      # twin-Q first layer: the small-output pair, x shared, db folded
                    (512, 1024, 50, 1, False, True),     # actor first layer
                    (96, 40, 24, 1, False, False),       # neither family fits both: two launches
                    (256, 1024, 1024, 2, False, False)]


@pytest.mark.parametrize("mfma", ["auto", "f32"])
@pytest.mark.parametrize("B,N,K,nb,shared_x,with_db", LINEAR_BWD_CASES)
def test_linear_bwd_pair_launch(ops, B, N, K, nb, shared_x, with_db, mfma):
    """curla_linear_bwd: dW = dy^T x (+ db) and dx = (dy W) masked from one launch, against float64 and -- on the f32-input
    MFMA, where the pair and the single launches take the same tiles -- against the two separate products bit for bit
    (same kernels, same tiles, same order of summation).  Under `auto` the pair counts its two products' 128 x 64 tiles
    together, so a product can be on bf16x3 in the pair and on the f32 form alone: float64 is the only arbiter there."""
    from curla_amd import _lib
    with _lib.option("gemm_mfma", mfma):
        _linear_bwd_pair(ops, B, N, K, nb, shared_x, with_db, bitwise=(mfma == "f32"))


def _linear_bwd_pair(ops, B, N, K, nb, shared_x, with_db, bitwise):
    dy, W = rnd(nb, B, N, seed=71), rnd(nb, N, K, seed=72) * 0.1
    x = rnd(1 if shared_x else nb, B, K, seed=73)
    mask = rnd(nb, B, K, seed=74)
    xe = x.expand(nb, B, K)
    dyd, Wd, xd, md = dev(dy), dev(W), dev(x), dev(mask)
    sx = 0 if shared_x else B * K
    dW = torch.full((nb, N, K), float("nan"), device="cuda")
    dx = torch.full((nb, B, K), float("nan"), device="cuda")
    db = torch.full((nb, N), float("nan"), device="cuda") if with_db else None
    if with_db:
        assert ops.linear_dw_folds_bias(B, N, K, nb)
    ops.linear_bwd(dyd, B * N, xd, sx, Wd, N * K, dW, N * K, dx, B * K, B, N, K, nb, mask=md, smask=B * K, db=db, sdb=N)
    tol = 3e-5
    check(f"linear_bwd dW {B}x{N}x{K} nb{nb}", dW.cpu(), torch.einsum("zbn,zbk->znk", dy.double(), xe.double()).float(), tol)
    check(f"linear_bwd dx {B}x{N}x{K} nb{nb}", dx.cpu(),
          (torch.einsum("zbn,znk->zbk", dy.double(), W.double()) * (mask > 0)).float(), tol)
    if with_db:
        check(f"linear_bwd db {B}x{N}x{K} nb{nb}", db.cpu(), dy.double().sum(1).float(), tol)
    dW2, dx2 = torch.full_like(dW, float("nan")), torch.full_like(dx, float("nan"))
    db2 = torch.full_like(db, float("nan")) if with_db else None
    ops.linear_dw(dyd, B * N, xd, sx, dW2, N * K, B, N, K, nb, colsum=db2, s_colsum=N)
    ops.linear_dx(dyd, B * N, Wd, N * K, dx2, B * K, B, N, K, nb, mask=md, smask=B * K)
    if bitwise:
        assert torch.equal(dW, dW2) and torch.equal(dx, dx2)
    else:
        check(f"linear_bwd dW pair against single {B}x{N}x{K} nb{nb}", dW.cpu(), dW2.cpu(), tol)
        check(f"linear_bwd dx pair against single {B}x{N}x{K} nb{nb}", dx.cpu(), dx2.cpu(), tol)
    if with_db:
        assert torch.equal(db, db2)


@pytest.mark.parametrize("B,Fd,K", [(24, 50, 3456), (9, 130, 800), (5, 64, 288), (3, 256, 512), (515, 50, 64)])
def test_gemm_splitk_and_fc_ln(ops, B, Fd, K):
    h, W, bias = rnd(B, K, seed=31), rnd(Fd, K, seed=32, scale=0.05), rnd(Fd, seed=33)
    gamma, beta = 1 + 0.1 * rnd(Fd, seed=34), 0.1 * rnd(Fd, seed=35)
    ks = 7
    part = torch.full((ks, B, Fd), float("nan"), device="cuda")
    ops.gemm(dev(h), 0, K, 0, dev(W), 0, K, 0, part, Fd, 0, B, Fd, K, 1, ksplit=ks, split_stride=B * Fd)
    hl = h.clone().requires_grad_(True)
    Wl, bl = W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    gl, betal = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    fc = F.linear(hl, Wl, bl)
    y_ref = F.layer_norm(fc, (Fd,), gl, betal, 1e-5)
    y, fco, xhat = (torch.empty(B, Fd, device="cuda") for _ in range(3))
    rstd = torch.empty(B, device="cuda")
    act = rnd(B, 3, seed=38)
    xa = torch.full((B, Fd + 3), float("nan"), device="cuda")
    ops.fc_ln_fwd(part, ks, B * Fd, Fd, dev(bias), dev(gamma), dev(beta), B, Fd, y, fc_out=fco, xhat=xhat, rstd=rstd,
                  xa=xa, act=dev(act))
    check("fc splitk + bias", fco.cpu(), fc.detach())
    check("layernorm fwd", y.cpu(), y_ref.detach())
    assert torch.equal(xa.cpu(), torch.cat([y.cpu(), act], 1))  # torch.cat([z, action], 1) written by the same kernel
    dy = rnd(B, Fd, seed=36)
    y_ref.backward(dy)
    dx, dg, db_ = torch.empty(B, Fd, device="cuda"), torch.empty(Fd, device="cuda"), torch.empty(Fd, device="cuda")
    dbin = torch.empty(Fd, device="cuda")
    ops.ln_bwd(dev(dy), xhat, rstd, dev(gamma), B, Fd, dx, dgamma=dg, dbeta=db_, dbias_in=dbin)
    check("fc dbias fused into the LayerNorm backward", dbin.cpu(), bl.grad)
    check("layernorm dgamma", dg.cpu(), gl.grad)
    check("layernorm dbeta", db_.cpu(), betal.grad)
    fcl = fc.detach().clone().requires_grad_(True)
    (F.layer_norm(fcl, (Fd,), gamma, beta, 1e-5) * dy).sum().backward()
    check("layernorm dx", dx.cpu(), fcl.grad)
    # the incoming gradient as the sum of two strided row blocks (the twin halves of the Q-input gradient
    # [2][B][Fd + 3], torch.cat's backward, read in place)
    half = rnd(2, B, Fd + 3, seed=39)
    half[1, :, :Fd] = dy - half[0, :, :Fd]
    twin = dev(half)
    dx2, dg2, db2, dbin2 = (torch.full_like(t, float("nan")) for t in (dx, dg, db_, dbin))
    ops.ln_bwd(twin[0], xhat, rstd, dev(gamma), B, Fd, dx2, dgamma=dg2, dbeta=db2, dbias_in=dbin2, dy2=twin[1], ld=Fd + 3)
    check("twin ln_bwd dx", dx2.cpu(), fcl.grad, 2e-5)
    check("twin ln_bwd dgamma", dg2.cpu(), gl.grad, 2e-5)
    check("twin ln_bwd dbeta", db2.cpu(), betal.grad, 2e-5)
    check("twin ln_bwd fc dbias", dbin2.cpu(), bl.grad, 2e-5)
    # fc backward through the helpers used by the agent
    dW = torch.empty(Fd, K, device="cuda")
    ops.linear_dw(dx, 0, dev(h), 0, dW, 0, B, Fd, K)
    check("fc dW (TN gemm)", dW.cpu(), Wl.grad)
    dbias = torch.empty(Fd, device="cuda")
    ops.colsum(dx, B, Fd, Fd, 0, dbias, 0)
    check("fc dbias (colsum)", dbias.cpu(), bl.grad)
    dh = torch.empty(B, K, device="cuda")
    msk = rnd(B, K, seed=37)
    ops.linear_dx(dx, 0, dev(W), 0, dh, 0, B, Fd, K, mask=dev(msk))
    check("fc dh (NN gemm + relu mask)", dh.cpu(), hl.grad * (msk > 0))
    red = torch.empty(B, Fd, device="cuda")
    from curla_amd._lib import call, ptr, stream
    call("curla_splitk_reduce", ptr(part), ks, B * Fd, B, Fd, Fd, ptr(red), Fd, ptr(dev(bias)), 0, stream())
    check("splitk_reduce", red.cpu(), fc.detach())


def _head_ref(out2a, noise, lo, hi):
    A = noise.shape[1]
    mu, ls = out2a.chunk(2, dim=-1)
    t = torch.tanh(ls)
    log_std = lo + 0.5 * (hi - lo) * (t + 1)
    pi = mu + noise * log_std.exp()
    log_pi = (-0.5 * noise.pow(2) - log_std).sum(-1, keepdim=True) - 0.5 * np.log(2 * np.pi) * A
    mu_t, pi_t = torch.tanh(mu), torch.tanh(pi)
    log_pi = log_pi - torch.log(F.relu(1 - pi_t.pow(2)) + 1e-6).sum(-1, keepdim=True)
    return mu_t, pi_t, log_pi, log_std


@pytest.mark.parametrize("B,A", [(8, 2), (300, 2), (17, 3)])
def test_actor_head(ops, B, A):
    lo, hi = -10.0, 2.0
    out = rnd(B, 2 * A, seed=41).requires_grad_(True)
    noise = rnd(B, A, seed=42)
    mu_r, pi_r, lp_r, ls_r = _head_ref(out, noise, lo, hi)
    mu, pi, ls, tl = (torch.empty(B, A, device="cuda") for _ in range(4))
    lp = torch.empty(B, 1, device="cuda")
    ops.actor_head_fwd(dev(out.detach()), dev(noise), B, A, lo, hi, mu=mu, pi=pi, log_pi=lp, log_std=ls, tanh_ls=tl)
    for n, a, r in (("mu", mu, mu_r), ("pi", pi, pi_r), ("log_pi", lp, lp_r), ("log_std", ls, ls_r)):
        check(f"actor_head_fwd {n} B{B}", a.cpu(), r.detach())
    gpi = rnd(B, A, seed=43)
    log_alpha = torch.tensor(np.log(0.1))
    glp = float(log_alpha.exp()) / B
    ((pi_r * gpi).sum() + glp * lp_r.sum()).backward()
    dout = torch.empty(B, 2 * A, device="cuda")
    ops.actor_head_bwd(dev(gpi), dev(log_alpha), 1.0 / B, dev(noise), pi, ls, tl, B, A, lo, hi, dout)
    check(f"actor_head_bwd B{B}", dout.cpu(), out.grad)
    # d(loss)/d(pi) read in place from the twin-Q input gradient [2][B][F + A] (action columns summed over the twin)
    Fz = 5
    dxa = rnd(2, B, Fz + A, seed=44)
    dxa[1, :, Fz:] = gpi - dxa[0, :, Fz:]
    dout2 = torch.full_like(dout, float("nan"))
    ops.actor_head_bwd(None, dev(log_alpha), 1.0 / B, dev(noise), pi, ls, tl, B, A, lo, hi, dout2, twin_dxa=dev(dxa),
                       F=Fz)
    check(f"actor_head_bwd twin B{B}", dout2.cpu(), out.grad, 2e-5)
    # select_action form: no noise
    mu2 = torch.empty(B, A, device="cuda")
    ops.actor_head_fwd(dev(out.detach()), None, B, A, lo, hi, mu=mu2)
    check(f"actor_head_fwd mu-only B{B}", mu2.cpu(), mu_r.detach())
    # the head inside the trunk's last-layer launch (curla_mlp_out_head_fwd): same trunk output as the plain layer, same
    # head outputs as the head kernel fed with it, pi also written into the action columns of the Q input rows
    K = 64
    h, W, b = torch.relu(rnd(B, K, seed=45)).cuda(), (rnd(2 * A, K, seed=46) * 0.2).cuda(), rnd(2 * A, seed=47).cuda()
    o1 = torch.empty(B, 2 * A, device="cuda")
    ops.mlp_out_fwd(h, 0, W, 0, b, 0, o1, 0, B, 2 * A, K)
    r_mu, r_pi, r_ls, r_tl = (torch.empty(B, A, device="cuda") for _ in range(4))
    r_lp = torch.empty(B, 1, device="cuda")
    ops.actor_head_fwd(o1, dev(noise), B, A, lo, hi, mu=r_mu, pi=r_pi, log_pi=r_lp, log_std=r_ls, tanh_ls=r_tl)
    o2 = torch.full_like(o1, float("nan"))
    f_mu, f_pi, f_ls, f_tl = (torch.full((B, A), float("nan"), device="cuda") for _ in range(4))
    f_lp = torch.full((B, 1), float("nan"), device="cuda")
    xa = torch.full((B, Fz + A), float("nan"), device="cuda")
    ops.mlp_out_head_fwd(h, W, b, o2, dev(noise), B, A, K, lo, hi, mu=f_mu, pi=f_pi, log_pi=f_lp, log_std=f_ls,
                         tanh_ls=f_tl, xa=xa)
    assert torch.equal(o1, o2)
    for n, a_, r_ in (("mu", f_mu, r_mu), ("pi", f_pi, r_pi), ("log_std", f_ls, r_ls), ("tanh_ls", f_tl, r_tl)):
        assert torch.equal(a_, r_), n
    check(f"fused head log_pi B{B}", f_lp.cpu(), r_lp.cpu(), 1e-6)
    assert torch.equal(xa[:, Fz:], r_pi) and bool(torch.isnan(xa[:, :Fz]).all())


def test_losses(ops):
    B, A = 70, 2
    q = rnd(2, B, 1, seed=51).requires_grad_(True)
    tq, lp = rnd(2, B, 1, seed=52), rnd(B, 1, seed=53)
    r, nd = rnd(B, 1, seed=54), (rnd(B, 1, seed=55) > -1).float()
    log_alpha = torch.tensor(np.log(0.1), requires_grad=True)
    alpha = log_alpha.exp()
    target = (r + nd * 0.99 * (torch.min(tq[0], tq[1]) - alpha.detach() * lp)).float()
    tgt = torch.empty(B, 1, device="cuda")
    d_la = dev(log_alpha.detach())
    ops.td_target(dev(tq), B, dev(lp), dev(r), dev(nd), d_la, 0.99, B, tgt)
    check("td_target", tgt.cpu(), target)
    loss_ref = F.mse_loss(q[0], target) + F.mse_loss(q[1], target)
    loss_ref.backward()
    loss, dq = torch.empty(1, device="cuda"), torch.empty(2, B, 1, device="cuda")
    ops.critic_loss(dev(q.detach()), B, tgt, B, loss, dq)
    check("critic_loss", loss.cpu(), loss_ref.detach().reshape(1))
    check("critic_loss dq", dq.cpu(), q.grad)
    # the fused form used by update_critic: bit-identical to the two kernels above
    tgt2, loss2, dq3 = torch.empty(B, 1, device="cuda"), torch.empty(1, device="cuda"), torch.empty(2, B, 1, device="cuda")
    ops.critic_td_loss(dev(q.detach()), dev(tq), B, dev(lp), dev(r), dev(nd), d_la, 0.99, B, tgt2, loss2, dq3)
    assert torch.equal(tgt2, tgt) and torch.equal(loss2, loss) and torch.equal(dq3, dq)
    # actor / alpha
    q2 = rnd(2, B, 1, seed=56).requires_grad_(True)
    ls = rnd(B, A, seed=57)
    actor_loss = (alpha.detach() * lp - torch.min(q2[0], q2[1])).mean()
    actor_loss.backward()
    alpha_loss = (alpha * (-lp - (-2.0)).detach()).mean()
    alpha_loss.backward()
    ent = (0.5 * A * (1.0 + np.log(2 * np.pi)) + ls.sum(-1)).mean()
    sc, dq2 = torch.empty(4, device="cuda"), torch.empty(2, B, 1, device="cuda")
    dla = torch.zeros((), dtype=torch.float64, device="cuda")
    ops.actor_loss(dev(q2.detach()), B, dev(lp), dev(ls), A, d_la, -2.0, B, sc, dq2, dla)
    check("actor_loss scalars", sc.cpu(), torch.stack([actor_loss.detach(), alpha_loss.detach(), ent, alpha.detach()]).float())
    check("actor_loss dq", dq2.cpu(), q2.grad)
    check("dlog_alpha", dla.cpu(), log_alpha.grad)
    # both losses evaluated INSIDE the backward launch of the twin Q functions' last layer (curla_mlp_out_bwd_loss):
    # same dq / target bit for bit, same scalars up to the summation order, same layer gradients as mlp_out_bwd fed
    # with that dq
    K = 64
    h = torch.relu(rnd(2, B, K, seed=58)).cuda()
    S = K + 8
    Wf = torch.zeros(2, S, device="cuda")
    Wf[:, :K] = (rnd(2, K, seed=59) * 0.2).cuda()
    for kind, fields, ref_dq, ref_sc in (
            (1, dict(q=dev(q.detach()), target_q_twin=dev(tq), log_pi=dev(lp), reward=dev(r), not_done=dev(nd),
                     discount=0.99), dq3, loss2),
            (2, dict(q=dev(q2.detach()), log_pi=dev(lp), log_std=dev(ls), A=A, target_entropy=-2.0), dq2, sc)):
        dh_r, dW_r = torch.empty(2, B, K, device="cuda"), torch.zeros(2, S, device="cuda")
        gb_r = torch.full((2, S), float("nan"), device="cuda")
        ops.mlp_out_bwd(ref_dq, B, h, B * K, Wf, S, dh_r, B * K, dW_r, S, B, 1, K, 2, db_out=gb_r, db_hidden=gb_r[0, 4:], sdb=S)
        dh_f, dW_f = torch.full_like(dh_r, float("nan")), torch.zeros(2, S, device="cuda")
        gb_f = torch.full((2, S), float("nan"), device="cuda")
        dq_f, sc_f = torch.full((2, B, 1), float("nan"), device="cuda"), torch.full((4,), float("nan"), device="cuda")
        tgt_f = torch.full((B, 1), float("nan"), device="cuda")
        dla_f = torch.zeros((), dtype=torch.float64, device="cuda")
        ops.mlp_out_bwd_loss(dict(kind=kind, twin_stride=B, log_alpha=d_la, scalars=sc_f, dq=dq_f, target_q=tgt_f,
                                  dlog_alpha=dla_f, **fields),
                             h, B * K, Wf, S, dh_f, B * K, dW_f, S, B, K, db_out=gb_f, db_hidden=gb_f[0, 4:], sdb=S)
        assert torch.equal(dq_f, ref_dq), kind
        assert torch.equal(dh_f, dh_r) and torch.equal(dW_f, dW_r) and torch.equal(gb_f[:, :4 + K].nan_to_num(7.0), gb_r[:, :4 + K].nan_to_num(7.0))
        n_sc = 1 if kind == 1 else 4
        check(f"loss-in-backward scalars kind {kind}", sc_f[:n_sc].cpu(), ref_sc[:n_sc].cpu(), 1e-6)
        if kind == 1:
            assert torch.equal(tgt_f, tgt2)
        else:
            check("loss-in-backward dlog_alpha", dla_f.cpu(), dla.cpu(), 1e-6)


@pytest.mark.parametrize("B", [8, 64, 200, 512])
def test_curl_ce(ops, B):
    logits = (rnd(B, B, seed=61) * 3).requires_grad_(True)
    ref = F.cross_entropy(logits - logits.max(1)[0][:, None], torch.arange(B))
    ref.backward()
    rl, loss, dl = torch.empty(B, device="cuda"), torch.empty(1, device="cuda"), torch.empty(B, B, device="cuda")
    ops.curl_ce(dev(logits.detach()), B, B, rl, loss, dl)
    dl2 = torch.empty(B, B, device="cuda")
    ops.curl_ce(dev(logits.detach()), B, B, rl, None, dl2)  # the mean is optional (only computed when logged)
    assert torch.equal(dl, dl2)
    check(f"curl_ce loss B{B}", loss.cpu(), ref.detach().reshape(1))
    check(f"curl_ce dlogits B{B}", dl.cpu(), logits.grad)


@pytest.mark.parametrize("B,Fd,K", [(128, 50, 64), (512, 50, 196), (256, 49, 128), (1024, 52, 64), (384, 51, 64),
                                    # every B / 128 from 1 to 8: 5 and 7 walk phase B in runs of 4 k-steps (B / 32 = 20, 28)
                                    (640, 50, 64), (768, 50, 64), (896, 52, 64)])
def test_curl_head_one_launch(ops, B, Fd, K):
    """curla_curl_head + curla_fc_bwd_ln2: the CURL phase from (z_a, z_pos, W) to the loss, d(loss)/d(fc output) of the
    anchor encoder, the LayerNorm / fc-bias gradients and dW -- against autograd through the reference's formulas
    (curl_sac.py:211-222, 411-413: logits = z_a W z_pos^T, minus the row max, cross-entropy against arange(B)) with the
    anchor features as a LayerNorm output, as in the encoder (encoder.py:101)."""
    assert ops.curl_head_supported(B, Fd)
    fc = rnd(B, Fd, seed=201).requires_grad_(True)          # pre-LayerNorm features of the anchors
    gamma = (1 + 0.1 * rnd(Fd, seed=202)).requires_grad_(True)
    beta = (0.1 * rnd(Fd, seed=203)).requires_grad_(True)
    z_pos = rnd(B, Fd, seed=204) * 0.7
    W = torch.rand(Fd, Fd, generator=torch.Generator().manual_seed(205)).requires_grad_(True)
    z_a = F.layer_norm(fc, (Fd,), gamma, beta, 1e-5)
    logits = z_a @ (W @ z_pos.t())
    loss = F.cross_entropy(logits - logits.max(1)[0][:, None], torch.arange(B))
    loss.backward()
    with torch.no_grad():
        mean = fc.mean(1, keepdim=True)
        rstd_ref = (fc.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        xhat_ref = (fc - mean) * rstd_ref
    f = lambda *sh: torch.full(sh, float("nan"), device="cuda")  # noqa: E731
    za_d, zp_d, W_d = dev(z_a.detach()), dev(z_pos), dev(W.detach())
    wz = f(B, Fd)
    ops.linear_fwd(zp_d, 0, W_d, 0, None, 0, wz, 0, B, Fd, Fd)
    row_loss, lsum, dfc, lg, dlg, dz = f(B), f(1), f(B, Fd), f(B, B), f(B, B), f(B, Fd)
    lnp, wp = f(ops.ln_partial_floats(B, Fd)), f((B // 16) * Fd * Fd)
    dgam, dbet, dbias, dW = f(Fd), f(Fd), f(Fd), f(Fd, Fd)
    tok = ops.curl_head(za_d, zp_d, wz, dev(xhat_ref), dev(rstd_ref.flatten()), dev(gamma.detach()), B, Fd, row_loss, dfc,
                        lnp, wp, dgam, dbet, dbias, dW, loss=lsum, logits=lg, dlogits=dlg, dz=dz)
    assert bool(torch.isnan(dgam).all()) and bool(torch.isnan(dW).all())  # finished by the fc backward
    # the fc backward that follows in the agent (any activations will do here)
    Wfc, x = (rnd(Fd, K, seed=206) * 0.1).cuda(), torch.relu(rnd(B, K, seed=207)).cuda()
    gx, dwfc = f(B, K), f(Fd, K)
    ops.fc_bwd(dfc, Wfc, x, gx, dwfc, B, Fd, K, ln=tok)
    check(f"curl_head logits B{B} F{Fd}", lg.cpu(), logits.detach(), 2e-6)
    check(f"curl_head loss B{B}", lsum.cpu(), loss.detach().reshape(1))
    z_a.retain_grad()
    check(f"curl_head d fc_out B{B}", dfc.cpu(), fc.grad, 2e-5)
    check(f"curl_head dgamma B{B}", dgam.cpu(), gamma.grad, 2e-5)
    check(f"curl_head dbeta B{B}", dbet.cpu(), beta.grad, 2e-5)
    check(f"curl_head fc dbias B{B}", dbias.cpu(), fc.grad.sum(0), 2e-5)
    check(f"curl_head dW B{B}", dW.cpu(), W.grad, 2e-5)
    # the separate launches compute the same intermediate quantities
    lg2, rl2, dl2, dz2 = f(B, B), f(B), f(B, B), f(B, Fd)
    ops.linear_fwd(za_d, 0, wz, 0, None, 0, lg2, 0, B, B, Fd)
    ops.curl_ce(lg2, B, B, rl2, None, dl2)
    ops.linear_dx(dl2, 0, wz, 0, dz2, 0, B, B, Fd)
    check("curl_head row losses vs curl_ce", row_loss.cpu(), rl2.cpu(), 2e-6)
    check("curl_head dlogits vs curl_ce", dlg.cpu(), dl2.cpu(), 2e-5)
    check("curl_head dz vs the separate product", dz.cpu(), dz2.cpu(), 2e-5)
    # fc products untouched by the extra reduction work, and everything reproducible
    gx2, dwfc2 = f(B, K), f(Fd, K)
    ops.fc_bwd(dfc, Wfc, x, gx2, dwfc2, B, Fd, K)
    assert torch.equal(gx, gx2) and torch.equal(dwfc, dwfc2)
    dfc3, dW3, dgam3 = f(B, Fd), f(Fd, Fd), f(Fd)
    tok3 = ops.curl_head(za_d, zp_d, wz, dev(xhat_ref), dev(rstd_ref.flatten()), dev(gamma.detach()), B, Fd, f(B), dfc3,
                         lnp, wp, dgam3, f(Fd), f(Fd), dW3)
    ops.fc_bwd(dfc3, Wfc, x, f(B, K), f(Fd, K), B, Fd, K, ln=tok3)
    assert torch.equal(dfc3, dfc) and torch.equal(dW3, dW) and torch.equal(dgam3, dgam)


def test_concat_split_softupdate_mean(ops):
    B, Fd, A = 9, 50, 2
    z, a = rnd(B, Fd, seed=71), rnd(B, A, seed=72)
    xa = torch.empty(B, Fd + A, device="cuda")
    ops.concat(dev(z), dev(a), B, Fd, A, xa)
    assert torch.equal(xa.cpu(), torch.cat([z, a], 1))
    dxa = rnd(2, B, Fd + A, seed=73)
    dz, da = torch.empty(B, Fd, device="cuda"), torch.empty(B, A, device="cuda")
    ops.split_sum(dev(dxa), B * (Fd + A), B, Fd, A, dz=dz, dact=da)
    check("split_sum dz", dz.cpu(), (dxa[0] + dxa[1])[:, :Fd])
    check("split_sum dact", da.cpu(), (dxa[0] + dxa[1])[:, Fd:])
    p, t = rnd(100003, seed=74), rnd(100003, seed=75)
    td = dev(t)
    ops.soft_update(dev(p), td, 0.05)
    check("soft_update", td.cpu(), 0.05 * p + (1 - 0.05) * t, 1e-6)
    td2 = dev(t)
    ops.soft_update2(dev(p), td2, 40000, 0.05, 0.01)  # two rates over one flat block
    check("soft_update2 head", td2[:40000].cpu(), 0.05 * p[:40000] + (1 - 0.05) * t[:40000], 1e-6)
    check("soft_update2 tail", td2[40000:].cpu(), 0.01 * p[40000:] + (1 - 0.01) * t[40000:], 1e-6)
    assert torch.equal(td2[:40000], td[:40000])
    sc = rnd(50, 5, seed=76)  # ring scalar rows: action(3) | reward | not_done
    idx = torch.tensor([7, 0, 49, 7, 13], dtype=torch.int64)
    ga, gr, gn = torch.empty(5, 3, device="cuda"), torch.empty(5, 1, device="cuda"), torch.empty(5, 1, device="cuda")
    ops.gather_transition_scalars(dev(sc), dev(idx), 5, 3, ga, gr, gn)
    assert torch.equal(ga.cpu(), sc[idx, :3]) and torch.equal(gr.cpu(), sc[idx, 3:4]) and torch.equal(gn.cpu(), sc[idx, 4:5])
    # the same gather fed from a pinned host block (indices first, anything after), which the launch also copies
    for B_, extra in ((5, 3), (300, 900), (2, 0)):
        host = torch.empty(B_ * 8 + extra * 8, dtype=torch.uint8).pin_memory()
        idx = torch.randint(0, 50, (B_,), generator=torch.Generator().manual_seed(B_))
        host[:B_ * 8].view(torch.int64).copy_(idx)
        host[B_ * 8:].copy_(torch.randint(0, 256, (extra * 8,), generator=torch.Generator().manual_seed(7)).to(torch.uint8))
        blk = torch.zeros_like(host, device="cuda")
        ga, gr, gn = (torch.empty(B_, n, device="cuda") for n in (3, 1, 1))
        ops.sample_stage(ops.host_device_pointer(host), blk, host.numel(), dev(sc), B_, 3, ga, gr, gn)
        torch.cuda.synchronize()
        assert torch.equal(blk.cpu(), host)
        assert torch.equal(ga.cpu(), sc[idx, :3]) and torch.equal(gr.cpu(), sc[idx, 3:4]) and torch.equal(gn.cpu(), sc[idx, 4:5])
    from curla_amd._lib import CurlaHipError
    with pytest.raises(CurlaHipError):
        ops.host_device_pointer(torch.empty(64, dtype=torch.uint8))  # pageable memory: no device address
    m = torch.empty(1, device="cuda")
    ops.mean(dev(p[:777]), 777, m)
    check("mean", m.cpu(), p[:777].mean().reshape(1), 1e-5)


def test_bad_arguments_fail_loudly(ops):
    from curla_amd._lib import CurlaHipError
    x = torch.zeros(2, 9, 9, 6, device="cuda")  # 6 channels: not a multiple of 4 (round 6: 16 / 64 / ... run the generic path)
    with pytest.raises(CurlaHipError):
        ops.conv_s1_fwd(x, torch.zeros(6, 6, 3, 3, device="cuda"), torch.zeros(6, device="cuda"),
                        torch.zeros(2, 7, 7, 6, device="cuda"))
    with pytest.raises(CurlaHipError):
        ops.ObsRef.from_tensor(torch.zeros(1, 9, 20, 20))  # CPU tensor


def test_full_size_batch_independence(ops):
    """BASELINE.json's full size (B=512, 84x84x9 ring -> 76x76 crop, 37/35/33/31 maps): a sample's result must
    not depend on what else is in the minibatch.  The conv stack, the fc split-K product and the data gradient
    of the full batch equal, bit for bit, the same kernels run on the two halves; weight gradients (whose
    fixed summation order follows the work split) agree to 1e-5; and a handful of samples are checked against
    the PyTorch statement of the layer."""
    B, C = 512, 9
    g = torch.Generator(device="cuda").manual_seed(7)
    store = torch.randint(0, 256, (1024 * 84 * 84 * C + 32,), dtype=torch.uint8, device="cuda", generator=g)
    ring = store[:1024 * 84 * 84 * C].view(1024, 84, 84, C)
    idx = torch.randint(0, 1024, (B,), device="cuda", generator=g)
    h1 = torch.randint(0, 9, (B,), device="cuda", generator=g).int()
    w1 = torch.randint(0, 9, (B,), device="cuda", generator=g).int()
    w0, b0 = dev(rnd(32, C, 3, 3, seed=1, scale=0.1)), dev(rnd(32, seed=2, scale=0.1))
    ws, bs = dev(rnd(32, 32, 3, 3, seed=3, scale=0.1)), dev(rnd(32, seed=4, scale=0.1))

    def stack(lo, hi):
        n = hi - lo
        ref = ops.ObsRef.from_ring(ring, idx[lo:hi].contiguous(), h1[lo:hi].contiguous(), w1[lo:hi].contiguous(), n, (76, 76))
        a1 = torch.empty(n, 37, 37, 32, device="cuda")
        ops.conv1_fwd(ref, w0, b0, a1)
        acts = [a1]
        for hw in (35, 33, 31):
            o = torch.empty(n, hw, hw, 32, device="cuda")
            ops.conv_s1_fwd(acts[-1], ws, bs, o)
            acts.append(o)
        return ref, acts

    ref_full, full = stack(0, B)
    _, lo = stack(0, B // 2)
    _, hi = stack(B // 2, B)
    for l, (f, a, b) in enumerate(zip(full, lo, hi)):
        assert torch.equal(f[:B // 2], a) and torch.equal(f[B // 2:], b), f"conv layer {l + 1} depends on the batch"
    # a few samples against PyTorch (crop -> /255 -> conv stack)
    pick = [0, 255, 511]
    x = torch.stack([ring[idx[i], h1[i]:h1[i] + 76, w1[i]:w1[i] + 76] for i in pick]).permute(0, 3, 1, 2).float().cpu() / 255.0
    r = torch.relu(F.conv2d(x, w0.cpu(), b0.cpu(), stride=2))
    for _ in range(3):
        r = torch.relu(F.conv2d(r, ws.cpu(), bs.cpu()))
    check("full-size conv stack (3 samples of 512)", nchw(full[3][pick]), r)

    # fc split-K product: rows are independent
    K, Fd = 31 * 31 * 32, 50
    Wfc = dev(rnd(Fd, K, seed=5, scale=0.01))
    hflat = full[3].view(B, K)
    part = torch.empty(32, B, Fd, device="cuda")
    ops.gemm(hflat, 0, K, 0, Wfc, 0, K, 0, part, Fd, 0, B, Fd, K, 1, ksplit=32, split_stride=B * Fd)
    part_lo = torch.empty(32, B // 2, Fd, device="cuda")
    ops.gemm(lo[3].view(B // 2, K), 0, K, 0, Wfc, 0, K, 0, part_lo, Fd, 0, B // 2, Fd, K, 1, ksplit=32,
             split_stride=(B // 2) * Fd)
    assert torch.equal(part[:, :B // 2], part_lo), "fc product of a row depends on the batch"

    # backward of the last conv layer
    gy = torch.randn(B, 31, 31, 32, device="cuda", generator=g) * (full[3] > 0)
    gin = torch.empty(B, 33, 33, 32, device="cuda")
    ops.conv_s1_dgrad(gy, ws, full[2], gin)
    gin_lo = torch.empty(B // 2, 33, 33, 32, device="cuda")
    ops.conv_s1_dgrad(gy[:B // 2].contiguous(), ws, lo[2], gin_lo)
    assert torch.equal(gin[:B // 2], gin_lo), "dgrad of a sample depends on the batch"
    wsp = torch.empty(ops.wgrad_workspace_floats(32), device="cuda")
    dw, db = torch.empty(32, 32, 3, 3, device="cuda"), torch.empty(32, device="cuda")
    dwa, dba, dwb, dbb = torch.empty_like(dw), torch.empty_like(db), torch.empty_like(dw), torch.empty_like(db)
    ops.conv_s1_wgrad(full[2], gy, dw, db, wsp)
    ops.conv_s1_wgrad(lo[2], gy[:B // 2].contiguous(), dwa, dba, wsp)
    ops.conv_s1_wgrad(hi[2], gy[B // 2:].contiguous(), dwb, dbb, wsp)
    check("full-size wgrad = sum over halves", dw.cpu(), (dwa + dwb).cpu(), 1e-5)
    check("full-size bias grad = sum over halves", db.cpu(), (dba + dbb).cpu(), 1e-5)
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    ops.conv_s1_wgrad(full[2], gy, dw2, db2, wsp)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "wgrad is not run-to-run reproducible"
    # first-layer weight gradient from the ring
    g1 = torch.randn(B, 37, 37, 32, device="cuda", generator=g)
    ws1 = torch.empty(ops.wgrad_workspace_floats(C), device="cuda")
    dw1, db1 = torch.empty(32, C, 3, 3, device="cuda"), torch.empty(32, device="cuda")
    ops.conv1_wgrad(ref_full, g1, dw1, db1, ws1)
    xs = torch.stack([ring[idx[i], h1[i]:h1[i] + 76, w1[i]:w1[i] + 76] for i in range(0, B, 64)]).permute(0, 3, 1, 2).float() / 255.0
    # linearity spot check: gradient of 8 samples alone, against autograd
    sel = list(range(0, B, 64))
    sub = ops.ObsRef.from_ring(ring, idx[sel].contiguous(), h1[sel].contiguous(), w1[sel].contiguous(), len(sel), (76, 76))
    dws, dbs = torch.empty_like(dw1), torch.empty_like(db1)
    ops.conv1_wgrad(sub, g1[sel].contiguous(), dws, dbs, ws1)
    wref = w0.cpu().clone().requires_grad_(True)
    F.conv2d(xs.cpu(), wref, None, stride=2).backward(g1[sel].permute(0, 3, 1, 2).cpu())
    check("conv1 wgrad from the ring (8 of 512)", dws.cpu(), wref.grad)
    assert torch.isfinite(dw1).all() and float(dw1.abs().max()) > 0


def test_colsum3(ops):
    """Three bias gradients of a batched MLP backward in one launch."""
    M, nb = 300, 2
    Ns = (1, 70, 33)
    Xs = [rnd(nb, M, n, seed=50 + i) for i, n in enumerate(Ns)]
    outs = [torch.full((nb, 128), float("nan"), device="cuda") for _ in Ns]
    ops.colsum3(dev(Xs[0]), Ns[0], dev(Xs[1]), Ns[1], dev(Xs[2]), Ns[2], M, outs[0], outs[1], outs[2], 128, nb)
    for i, n in enumerate(Ns):
        check(f"colsum3[{i}] N={n}", outs[i][:, :n].cpu(), Xs[i].sum(1))
        assert torch.isnan(outs[i][:, n:]).all()  # nothing written past N


@pytest.mark.parametrize("M,N,K,nb", [(512, 1, 1024, 2), (512, 4, 1024, 1), (37, 3, 100, 2), (5, 16, 64, 1), (130, 6, 96, 3)])
def test_mlp_out_layer(ops, M, N, K, nb):
    """Last layer of the trunks (hidden -> 1 or 2|A| outputs) as row dot products / one fused backward pass,
    batched over twin networks with a parameter stride, against plain matmuls."""
    h = torch.relu(rnd(nb, M, K, seed=1))
    W, b, dy = rnd(nb, N, K, seed=2, scale=0.2), rnd(nb, N, seed=3), rnd(nb, M, N, seed=4)
    pad = 8  # twin parameter blocks are `stride` floats apart, not dense
    Wd = torch.zeros(nb, N * K + pad, device="cuda")
    Wd[:, :N * K] = dev(W).view(nb, -1)
    bd = torch.zeros(nb, N + 4, device="cuda")
    bd[:, :N] = dev(b)
    out = torch.full((nb, M, N), float("nan"), device="cuda")
    ops.mlp_out_fwd(dev(h), M * K, Wd, N * K + pad, bd, N + 4, out, M * N, M, N, K, nb)
    check(f"mlp_out fwd {M}x{N}x{K} nb{nb}", out.cpu(), torch.einsum("zmk,znk->zmn", h, W) + b[:, None, :])
    dh = torch.full((nb, M, K), float("nan"), device="cuda")
    dWd = torch.full((nb, N * K + pad), float("nan"), device="cuda")
    ops.mlp_out_bwd(dev(dy), M * N, dev(h), M * K, Wd, N * K + pad, dh, M * K, dWd, N * K + pad, M, N, K, nb)
    check(f"mlp_out dh {M}x{N}x{K} nb{nb}", dh.cpu(), torch.einsum("zmn,znk->zmk", dy, W) * (h > 0))
    check(f"mlp_out dW {M}x{N}x{K} nb{nb}", dWd[:, :N * K].view(nb, N, K).cpu(), torch.einsum("zmn,zmk->znk", dy, h))
    assert bool(torch.isnan(dWd[:, N * K:]).all())  # nothing written between the blocks
    dh2 = torch.full((nb, M, K), float("nan"), device="cuda")
    ops.mlp_out_bwd(dev(dy), M * N, dev(h), M * K, Wd, N * K + pad, dh2, M * K, None, 0, M, N, K, nb)  # data gradient only
    assert torch.equal(dh, dh2)
    # ... and with the two bias gradients (column sums of dy and of dh) from the same launch
    dh3, dW3 = torch.full_like(dh, float("nan")), torch.full_like(dWd, float("nan"))
    S = 32 + K + 7  # both live in one gradient block per twin, `S` floats apart: db_out at 0, db_hidden at 32
    gb = torch.full((nb, S), float("nan"), device="cuda")
    ops.mlp_out_bwd(dev(dy), M * N, dev(h), M * K, Wd, N * K + pad, dh3, M * K, dW3, N * K + pad, M, N, K, nb,
                    db_out=gb, db_hidden=gb[0, 32:], sdb=S)
    assert torch.equal(dh3, dh) and torch.equal(dW3[:, :N * K], dWd[:, :N * K])
    check(f"mlp_out db_out {M}x{N}x{K} nb{nb}", gb[:, :N].cpu(), dy.sum(1))
    check(f"mlp_out db_hidden {M}x{N}x{K} nb{nb}", gb[:, 32:32 + K].cpu(), dh.cpu().sum(1))
    assert bool(torch.isnan(gb[:, N:32]).all()) and bool(torch.isnan(gb[:, 32 + K:]).all())


def _flat_params(sizes, device, seed=0):
    """Parameters laid out like the agent's flat buffers: 4-float aligned slots, .grad views of a mirror buffer."""
    offs, off = [], 0
    for n in sizes:
        offs.append(off)
        off += (int(np.prod(n)) + 3) & ~3
    g = torch.Generator().manual_seed(seed)
    flat = torch.zeros(off, device=device)
    gflat = torch.zeros(off, device=device)
    params = []
    for n, o in zip(sizes, offs):
        cnt = int(np.prod(n))
        flat[o:o + cnt] = torch.randn(cnt, generator=g).to(device)
        p = torch.nn.Parameter(flat[o:o + cnt].view(n))
        p.grad = gflat[o:o + cnt].view(n)
        params.append(p)
    return flat, gflat, params


@pytest.mark.parametrize("betas", [(0.9, 0.999), (0.5, 0.999)])
def test_flat_adam_matches_torch_adam(betas):
    """curla_adam_step (optim.FlatAdam) against torch.optim.Adam on the CPU (the reference's optimizer,
    curl_sac.py:299-313) over 6 steps; one parameter loses its gradient for two steps in the middle (the
    detach_encoder case: Adam must skip it and keep its own step count) and state_dict() round-trips into a fresh
    optimizer."""
    from curla_amd.optim import FlatAdam
    sizes = [(32, 9, 3, 3), (32,), (50, 1203), (50,), (1,), (1024, 57), (7,)]
    flat, gflat, params = _flat_params(sizes, "cuda")
    ref_params = [torch.nn.Parameter(p.detach().cpu().clone()) for p in params]
    opt = FlatAdam(params, flat, gflat, lr=1e-3, betas=betas)
    ref = torch.optim.Adam(ref_params, lr=1e-3, betas=betas)
    gen = torch.Generator().manual_seed(5)
    saved = params[2].grad
    for step in range(6):
        skip = step in (2, 3)
        for i, (p, r) in enumerate(zip(params, ref_params)):
            gr = torch.randn(p.shape, generator=gen) * (10.0 ** (i % 3 - 2))
            if i == 2:
                p.grad = None if skip else saved
                r.grad = None if skip else gr
                if skip:
                    continue
            else:
                r.grad = gr
            p.grad.copy_(gr.cuda())
        opt.step()
        ref.step()
        if step == 3:  # a resumed optimizer continues identically
            sd = opt.state_dict()
            opt = FlatAdam(params, flat, gflat, lr=1e-3, betas=betas)
            opt.load_state_dict(sd)
    torch.cuda.synchronize()
    for i, (p, r) in enumerate(zip(params, ref_params)):
        check(f"flat_adam{betas} param{i}", p.detach(), r.detach(), 2e-6)
        st, rst = opt.state[p], ref.state[r]
        check(f"flat_adam{betas} exp_avg{i}", st["exp_avg"], rst["exp_avg"], 2e-6)
        check(f"flat_adam{betas} exp_avg_sq{i}", st["exp_avg_sq"], rst["exp_avg_sq"], 2e-6)
    sd, rsd = opt.state_dict(), ref.state_dict()
    assert sd["param_groups"][0]["betas"] == rsd["param_groups"][0]["betas"]
    assert set(sd["state"]) == set(rsd["state"])
    for k in sd["state"]:
        assert set(sd["state"][k]) == set(rsd["state"][k])
        assert float(sd["state"][k]["step"]) == float(rsd["state"][k]["step"])
    # padding between the slots never moves
    pad = 32 * 81 + 32 + 50 * 1203  # the (50, 1203) slot is 60150 floats + 2 of padding
    assert float(flat[pad:pad + 2].abs().sum()) == 0.0


@pytest.mark.parametrize("B,Fd,K", [(24, 50, 3456), (5, 50, 288), (3, 64, 512), (9, 13, 100), (515, 50, 196), (64, 7, 64),
                                    (130, 33, 1028)])
def test_fc_backward_streaming(ops, B, Fd, K):
    """curla_fc_dx / curla_fc_dw (the encoder fc layer's backward products, streamed straight into MFMA registers)
    against PyTorch: ragged batch, feature counts that are not multiples of 4 or 16, a last 64-column block that is
    only partly there, with and without the ReLU mask."""
    assert ops.fc_bwd_streams(Fd, K)
    dz, W, x = rnd(B, Fd, seed=51), rnd(Fd, K, seed=52, scale=0.1), rnd(B, K, seed=53)
    ref_dx = dz @ W
    out = torch.full((B, K), float("nan"), device="cuda")
    ops.fc_dx(dev(dz), dev(W), out, B, Fd, K, mask=dev(x))
    check(f"fc_dx masked {B}x{Fd}x{K}", out.cpu(), ref_dx * (x > 0))
    out2 = torch.full((B, K), float("nan"), device="cuda")
    ops.fc_dx(dev(dz), dev(W), out2, B, Fd, K)
    check(f"fc_dx {B}x{Fd}x{K}", out2.cpu(), ref_dx)
    dW = torch.full((Fd, K), float("nan"), device="cuda")
    ops.fc_dw(dev(dz), dev(x), dW, B, Fd, K)
    check(f"fc_dw {B}x{Fd}x{K}", dW.cpu(), dz.t() @ x)
    # the generic GEMM route gives the same numbers to rounding
    dW2 = torch.empty(Fd, K, device="cuda")
    ops.linear_dw(dev(dz), 0, dev(x), 0, dW2, 0, B, Fd, K)
    check(f"fc_dw vs gemm {B}x{Fd}x{K}", dW.cpu(), dW2.cpu(), 2e-5)


@pytest.mark.parametrize("B1,B2,H,W", [(5, 3, 13, 16), (2, 7, 37, 37), (600, 300, 9, 9), (1, 1, 35, 35)])
def test_conv_s1_fwd_two_problems(ops, s1_impl, B1, B2, H, W):
    """curla_conv3x3_s1_fwd2: two minibatches, each with its own weights, in one launch -- bit-identical to the two
    single launches (a workgroup walks into the second problem and re-builds its weight registers there)."""
    x1, x2 = rnd(B1, H, W, 32, seed=61).cuda(), rnd(B2, H, W, 32, seed=62).cuda()
    w1, w2 = (rnd(32, 32, 3, 3, seed=63) * 0.1).cuda(), (rnd(32, 32, 3, 3, seed=64) * 0.1).cuda()
    b1, b2 = (rnd(32, seed=65) * 0.1).cuda(), (rnd(32, seed=66) * 0.1).cuda()
    r1, r2 = torch.empty(B1, H - 2, W - 2, 32, device="cuda"), torch.empty(B2, H - 2, W - 2, 32, device="cuda")
    ops.conv_s1_fwd(x1, w1, b1, r1)
    ops.conv_s1_fwd(x2, w2, b2, r2)
    o1, o2 = torch.full_like(r1, float("nan")), torch.full_like(r2, float("nan"))
    ops.conv_s1_fwd2(x1, w1, b1, o1, x2, w2, b2, o2)
    assert torch.equal(o1, r1) and torch.equal(o2, r2)


@pytest.mark.parametrize("B1,B2,C", [(6, 3, 9), (3, 5, 12)])
def test_conv1_fwd_two_problems(ops, u8_impl, B1, B2, C):
    """curla_conv1_fwd2: two gathers from one uint8 ring with their own indices, crop offsets and weights."""
    Hs, Ws, Hc, Wc = 30, 34, 26, 28
    g = torch.Generator().manual_seed(71)
    n = 11
    store = torch.randint(0, 256, (n * Hs * Ws * C + 32,), dtype=torch.uint8, generator=g).cuda()
    ring = store[:n * Hs * Ws * C].view(n, Hs, Ws, C)
    refs, ws, bs, outs = [], [], [], []
    for k, B in enumerate((B1, B2)):
        idx = torch.randint(0, n, (B,), generator=g).cuda()
        h1 = torch.randint(0, Hs - Hc + 1, (B,), generator=g, dtype=torch.int32).cuda()
        w1 = torch.randint(0, Ws - Wc + 1, (B,), generator=g, dtype=torch.int32).cuda()
        refs.append(ops.ObsRef.from_ring(ring, idx, h1, w1, B, (Hc, Wc)))
        ws.append((rnd(32, C, 3, 3, seed=72 + k) * 0.1).cuda())
        bs.append((rnd(32, seed=74 + k) * 0.1).cuda())
        outs.append(torch.empty(B, (Hc - 3) // 2 + 1, (Wc - 3) // 2 + 1, 32, device="cuda"))
        ops.conv1_fwd(refs[k], ws[k], bs[k], outs[k])
    assert ops.conv1_pairable(refs[0], refs[1])
    o1, o2 = torch.full_like(outs[0], float("nan")), torch.full_like(outs[1], float("nan"))
    ops.conv1_fwd2(refs[0], ws[0], bs[0], o1, refs[1], ws[1], bs[1], o2)
    assert torch.equal(o1, outs[0]) and torch.equal(o2, outs[1])


@pytest.mark.parametrize("B,Fd,K,nprob,ns,blocked", [(128, 50, 3456, 1, 7, False), (512, 50, 30752, 3, 43, True),
                                                     (256, 64, 640, 2, 20, False), (128, 58, 96, 4, 3, True),
                                                     (384, 50, 1184, 2, 37, True), (128, 50, 416, 1, 13, False)])
def test_fc_forward_streaming_kernel(ops, B, Fd, K, nprob, ns, blocked):
    """curla_fc_fwd_multi: the encoder fc layer's forward as split-K partial sums (three feature tiles on the matrix pipe
    + two features on FMAs at F = 50; W through LDS, x straight from memory), up to four (x, W) pairs per launch, x
    row-major or in the blocked layout: the sum over the splits against x @ W^T in float64, single splits against the k
    ranges they stand for, nothing written around the outputs, reproducible run to run."""
    assert ops.fc_fwd_supported(B, Fd, K)
    xs = [rnd(B, K, seed=101 + i).cuda() for i in range(nprob)]
    xin = [ops.to_blocked(x, B, K) for x in xs] if blocked else xs
    if blocked:
        assert torch.equal(ops.from_blocked(xin[0], B, K), xs[0])
    Ws = [(rnd(Fd, K, seed=111 + i) * 0.05).cuda() for i in range(nprob)]
    guard = 64
    bufs = [torch.full((ns * B * Fd + 2 * guard,), float("nan"), device="cuda") for _ in range(nprob)]
    outs = [b[guard:guard + ns * B * Fd].view(ns, B, Fd) for b in bufs]
    ops.fc_fwd_multi(xin, Ws, outs, B, Fd, K, ns, B * Fd, blocked=blocked)
    nsl = K // 32
    for i in range(nprob):
        ref = xs[i].cpu().double() @ Ws[i].cpu().double().t()
        check(f"fc_fwd_multi sum of splits {B}x{Fd}x{K} problem {i}", outs[i].double().sum(0).cpu(), ref, 2e-6)
        for s_ in (0, ns // 2, ns - 1):
            k0, k1 = 32 * (nsl * s_ // ns), 32 * (nsl * (s_ + 1) // ns)
            ref_s = xs[i][:, k0:k1].cpu().double() @ Ws[i][:, k0:k1].cpu().double().t()
            check(f"fc_fwd_multi split {s_} of {ns}", outs[i][s_].cpu(), ref_s, 2e-6)
        assert bool(torch.isnan(bufs[i][:guard]).all()) and bool(torch.isnan(bufs[i][-guard:]).all())
    again = [torch.empty_like(o) for o in outs]
    ops.fc_fwd_multi(xin, Ws, again, B, Fd, K, ns, B * Fd, blocked=blocked)
    for a, o in zip(again, outs):
        assert torch.equal(a, o)


def test_gemm_multi_matches_single_launches(ops):
    """curla_gemm_multi: three unrelated split-K products of one shape, operands by pointer, in one launch --
    bit-identical to three curla_gemm launches."""
    B, Fd, K, ks = 24, 50, 3456, 6
    hs = [rnd(B, K, seed=81 + i).cuda() for i in range(3)]
    Ws = [(rnd(Fd, K, seed=84 + i) * 0.05).cuda() for i in range(3)]
    refs = [torch.empty(ks, B, Fd, device="cuda") for _ in range(3)]
    for h, W, r in zip(hs, Ws, refs):
        ops.gemm(h, 0, K, 0, W, 0, K, 0, r, Fd, 0, B, Fd, K, 1, ksplit=ks, split_stride=B * Fd)
    outs = [torch.full((ks, B, Fd), float("nan"), device="cuda") for _ in range(3)]
    ops.gemm_multi(hs, Ws, outs, B, Fd, K, ksplit=ks, split_stride=B * Fd)
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)
    check("gemm_multi vs torch", outs[1].sum(0).cpu(), hs[1].cpu() @ Ws[1].cpu().t())


@pytest.mark.parametrize("B,din,H", [(24, 53, 64), (130, 54, 128)])
def test_mlp_forward_two_level_batch(ops, B, din, H):
    """curla_gemm_nested / curla_mlp_out_fwd_nested: the twin Q functions of two critics (parameters a fixed distance
    apart, inputs [2][B, din]) in one launch per layer -- bit-identical to one twin launch per critic."""
    from curla_amd.curl_sac import _mlp_fwd

    class P:
        pass
    blk = ((H * din + 3) & ~3) + H + H * H + H + ((H + 3) & ~3) + 4  # one Q function: W0 b0 W1 b1 W2 b2
    tot = 2 * blk + 8
    flat = (rnd(2 * tot, seed=301) * 0.1).cuda()

    def mlp(base):
        o, m = base, P()
        m.W, m.b, m.stride = [], [], blk
        for n_out, n_in in ((H, din), (H, H), (1, H)):
            m.W.append(flat[o:o + n_out * n_in].view(n_out, n_in))
            o += (n_out * n_in + 3) & ~3
            m.b.append(flat[o:o + n_out])
            o += (n_out + 3) & ~3
        return m
    x = rnd(2, B, din, seed=302).cuda()
    ref_h1, ref_h2, ref_q = (torch.empty(2, 2, B, d, device="cuda") for d in (H, H, 1))
    for o in range(2):
        _mlp_fwd(x[o], 0, mlp(o * tot), 2, B, din, H, 1, ref_h1[o], ref_h2[o], ref_q[o])
    h1, h2, q = (torch.full((2, 2, B, d), float("nan"), device="cuda") for d in (H, H, 1))
    _mlp_fwd(x, 0, mlp(0), 2, B, din, H, 1, h1, h2, q, outer=(2, tot))
    assert torch.equal(h1, ref_h1) and torch.equal(h2, ref_h2) and torch.equal(q, ref_q)
    m = mlp(tot + blk)  # the second critic's second twin against torch
    xc = x[1].cpu()
    want = torch.relu(torch.relu(xc @ m.W[0].cpu().t() + m.b[0].cpu()) @ m.W[1].cpu().t() + m.b[1].cpu()) @ m.W[2].cpu().t() + m.b[2].cpu()
    check("nested mlp vs torch", q[1, 1].cpu(), want)


@pytest.mark.parametrize("B,H,W", [(3, 13, 16), (2, 37, 37), (300, 9, 9), (5, 17, 15)])
def test_conv_s1_backward_one_launch(ops, s1_impl, s1_wgrad, B, H, W):
    """curla_conv3x3_s1_bwd_slabs: weight-gradient slabs and data gradient of a layer in one launch.  The data gradient
    is bit-identical to the separate kernel's; the weight gradient is dealt to as many or half as many workgroups
    (slabs) as the separate kernel's, i.e. the same sums in another fixed order."""
    x = torch.relu(rnd(B, H, W, 32, seed=91)).cuda()
    g = rnd(B, H - 2, W - 2, 32, seed=92).cuda()
    w = (rnd(32, 32, 3, 3, seed=93) * 0.1).cuda()
    ws1 = torch.zeros(ops.wgrad_workspace_floats(32), device="cuda")
    ws2 = torch.zeros_like(ws1)
    gin1, gin2 = torch.empty_like(x), torch.full_like(x, float("nan"))
    n1 = ops.conv_s1_wgrad_slabs(x, g, ws1)
    ops.conv_s1_dgrad(g, w, x, gin1)
    n2 = ops.conv_s1_bwd_slabs(x, g, w, gin2, ws2)
    assert n2 in (n1, min(n1, ops.cu_count())) and torch.equal(gin1, gin2)
    dw1, db1, dw2, db2 = (torch.empty(s, device="cuda") for s in ((32, 32, 3, 3), (32,), (32, 32, 3, 3), (32,)))
    ops.wgrad_reduce_multi([(ws1, n1, dw1, db1)])
    ops.wgrad_reduce_multi([(ws2, n2, dw2, db2)])
    if n1 == n2:
        assert torch.equal(dw1, dw2) and torch.equal(db1, db2)
    else:
        check(f"bwd_slabs dW {B}x{H}x{W}", dw2.cpu(), dw1.cpu(), 2e-6)
        check(f"bwd_slabs db {B}x{H}x{W}", db2.cpu(), db1.cpu(), 2e-6)
    ws3, gin3, dw3, db3 = torch.zeros_like(ws1), torch.empty_like(x), torch.empty_like(dw1), torch.empty_like(db1)
    n3 = ops.conv_s1_bwd_slabs(x, g, w, gin3, ws3)
    ops.wgrad_reduce_multi([(ws3, n3, dw3, db3)])
    assert n3 == n2 and torch.equal(dw3, dw2) and torch.equal(db3, db2)  # run-to-run reproducible


def test_policy_noise_drawn_inside_the_head_launch(ops):
    """rng=(seed, offset): the head draws its own standard-normal noise (Philox4x32-10 + Box-Muller, curla_hip.h) and
    stores it; everything downstream of the noise is bit-identical to the explicit-noise form fed with those numbers;
    the stream is a pure function of (seed, offset), consecutive calls with advanced offsets do not overlap, and the
    numbers look like N(0, 1)."""
    lo, hi = -10.0, 2.0
    B, A = 32768, 8
    out = rnd(B, 2 * A, seed=61).cuda()
    f = lambda *sh: torch.full(sh, float("nan"), device="cuda")  # noqa: E731

    def draw(seed, off, head="head"):
        nz, pi, lp = f(B, A), f(B, A), f(B, 1)
        if head == "head":
            ops.actor_head_fwd(out, nz, B, A, lo, hi, pi=pi, log_pi=lp, rng=(seed, off))
        else:  # inside the trunk's last-layer launch
            K = 64
            h, W, b = torch.relu(rnd(B, K, seed=62)).cuda(), (rnd(2 * A, K, seed=63) * 0.2).cuda(), rnd(2 * A, seed=64).cuda()
            ops.mlp_out_head_fwd(h, W, b, f(B, 2 * A), nz, B, A, K, lo, hi, pi=pi, log_pi=lp, rng=(seed, off))
        return nz, pi, lp
    n1, p1, l1 = draw(1234, 10)
    assert bool(torch.isfinite(n1).all())
    x = n1.double().flatten()
    m, sd = float(x.mean()), float(x.std())
    kurt, skew = float(((x - m) ** 4).mean() / sd ** 4), float(((x - m) ** 3).mean() / sd ** 3)
    assert abs(m) < 0.01 and abs(sd - 1) < 0.01 and abs(kurt - 3) < 0.06 and abs(skew) < 0.02, (m, sd, kurt, skew)
    assert float(x.abs().max()) > 4.0  # tails are there (262144 draws: P(no |x| > 4) ~ 1e-7)
    # neighbours in the stream are uncorrelated (Box-Muller pairs share a radius: cos / sin of one angle)
    assert abs(float((x[:-1] * x[1:]).mean())) < 0.01
    # explicit-noise form on the same numbers: identical outputs
    pi2, lp2 = f(B, A), f(B, 1)
    ops.actor_head_fwd(out, n1, B, A, lo, hi, pi=pi2, log_pi=lp2)
    assert torch.equal(p1, pi2) and torch.equal(l1, lp2)
    # a pure function of (seed, offset); the trunk-launch form draws the same stream
    n1b, _, _ = draw(1234, 10)
    n3, _, _ = draw(1234, 10, head="trunk")
    assert torch.equal(n1, n1b) and torch.equal(n1, n3)
    # next call's offset = + ceil(B A / 4): no number of the first call repeats at the same position or shifted by it
    n4, _, _ = draw(1234, 10 + B * A // 4)
    assert not bool((n4 == n1).any())
    n5, _, _ = draw(1234, 10 + 1)  # one counter further = the stream shifted by four numbers
    assert torch.equal(n5.flatten()[:-4], n1.flatten()[4:])
    n6, _, _ = draw(1235, 10)
    assert not bool((n6 == n1).any())


@pytest.mark.parametrize("B,Fd,K,twin", [(512, 50, 196, True), (32, 50, 3456, False), (515, 50, 196, False), (9, 13, 100, True),
                                          (24, 130, 64, False)])
def test_ln_param_grads_finished_inside_the_fc_backward(ops, B, Fd, K, twin):
    """ops.ln_bwd(..., defer=) + ops.fc_bwd / fc_dw(..., ln=): dx, the fc products and the LayerNorm / fc-bias gradients
    against the three-launch path (LayerNorm backward, its parameter-gradient launch, fc backward) -- dx and the fc
    products bit-identical, the column sums within rounding (another fixed order) and reproducible run to run."""
    if not ops.fc_bwd_streams(Fd, K):
        pytest.skip("the fc backward of this shape does not take the streaming kernels")
    gamma = (1 + 0.1 * rnd(Fd, seed=51)).cuda()
    xhat, rstd = rnd(B, Fd, seed=52).cuda(), (0.5 + rnd(B, seed=53).abs()).cuda()
    ld = Fd + 3 if twin else Fd
    dy = rnd(2, B, ld, seed=54).cuda()
    dy2 = dy[1] if twin else None
    W, x = (rnd(Fd, K, seed=55) * 0.1).cuda(), torch.relu(rnd(B, K, seed=56)).cuda()
    f = lambda *sh: torch.full(sh, float("nan"), device="cuda")  # noqa: E731
    for with_dx in (True, False):
        dx1, dg1, db1, dbi1, g1, dw1 = f(B, Fd), f(Fd), f(Fd), f(Fd), f(B, K), f(Fd, K)
        ops.ln_bwd(dy[0], xhat, rstd, gamma, B, Fd, dx1, dgamma=dg1, dbeta=db1, dbias_in=dbi1, dy2=dy2, ld=ld)
        if with_dx:
            ops.fc_bwd(dx1, W, x, g1, dw1, B, Fd, K)
        else:
            ops.fc_dw(dx1, x, dw1, B, Fd, K)
        res = []
        for _ in range(2):
            dx2, dg2, db2, dbi2, g2, dw2 = f(B, Fd), f(Fd), f(Fd), f(Fd), f(B, K), f(Fd, K)
            part = f(ops.ln_partial_floats(B, Fd))
            tok = ops.ln_bwd(dy[0], xhat, rstd, gamma, B, Fd, dx2, dgamma=dg2, dbeta=db2, dbias_in=dbi2, dy2=dy2, ld=ld,
                             defer=part)
            assert tok is not None and bool(torch.isnan(dg2).all())  # not finished yet
            if with_dx:
                ops.fc_bwd(dx2, W, x, g2, dw2, B, Fd, K, ln=tok)
            else:
                ops.fc_dw(dx2, x, dw2, B, Fd, K, ln=tok)
            res.append((dg2, db2, dbi2))
            assert torch.equal(dx1, dx2) and torch.equal(dw1, dw2) and (not with_dx or torch.equal(g1, g2))
            check(f"deferred ln dgamma {B}x{Fd}", dg2.cpu(), dg1.cpu(), 2e-6)
            check(f"deferred ln dbeta {B}x{Fd}", db2.cpu(), db1.cpu(), 2e-6)
            check(f"deferred fc dbias {B}x{Fd}", dbi2.cpu(), dbi1.cpu(), 2e-6)
        for a, b in zip(*res):
            assert torch.equal(a, b)


def test_fc_backward_one_launch(ops):
    """curla_fc_bwd: data + weight gradient of the encoder fc layer in ONE pass over the activations (the ReLU mask of
    the one and the column operand of the other are the same matrix).  The data gradient is bit-identical to
    curla_fc_dx; the weight gradient adds the batch rows in another (fixed) order than curla_fc_dw.  Shapes outside the
    instantiated one (49..52 features, whole 16-row tiles) fall back to the two launches."""
    for B, Fd, K, one_pass in ((32, 50, 3456, True), (512, 50, 196, True), (48, 52, 260, True), (64, 49, 64, True),
                               (24, 50, 3456, False), (515, 50, 196, False), (9, 13, 100, False)):
        dz, W = rnd(B, Fd, seed=95).cuda(), (rnd(Fd, K, seed=96) * 0.1).cuda()
        x = torch.relu(rnd(B, K, seed=97)).cuda()
        dx1, dw1 = torch.empty(B, K, device="cuda"), torch.empty(Fd, K, device="cuda")
        ops.fc_dx(dz, W, dx1, B, Fd, K, mask=x)
        ops.fc_dw(dz, x, dw1, B, Fd, K)
        guard = 64  # nothing may be written around the outputs
        bx, bw = torch.full((B * K + 2 * guard,), float("nan"), device="cuda"), torch.full((Fd * K + 2 * guard,), float("nan"), device="cuda")
        dx2, dw2 = bx[guard:guard + B * K].view(B, K), bw[guard:guard + Fd * K].view(Fd, K)
        ops.fc_bwd(dz, W, x, dx2, dw2, B, Fd, K)
        assert torch.equal(dx1, dx2)
        if one_pass:
            check(f"fc_bwd dW {B}x{Fd}x{K}", dw2.cpu(), dz.cpu().double().t().mm(x.cpu().double()).float(), 2e-5)
            dw3 = torch.empty_like(dw1)
            ops.fc_bwd(dz, W, x, torch.empty_like(dx1), dw3, B, Fd, K)
            assert torch.equal(dw2, dw3)  # fixed summation order
        else:
            assert torch.equal(dw1, dw2)
        for buf in (bx, bw):
            assert bool(torch.isnan(buf[:guard]).all()) and bool(torch.isnan(buf[-guard:]).all())


def test_flat_adam_step_pair_is_two_steps():
    """FlatAdam.step_pair(first, second) (the encoder stepped by encoder_optimizer and then by cpc_optimizer,
    curl_sac.py:418-423, in one pass) leaves exactly the parameters and moments of first.step(); second.step()."""
    from curla_amd.optim import FlatAdam
    sizes = [(50, 50), (32, 9, 3, 3), (32,), (50, 1203), (50,)]
    res = []
    for fused in (False, True):
        flat, gflat, params = _flat_params(sizes, "cuda", seed=3)
        enc = FlatAdam(params[1:], flat, gflat, lr=1e-3)
        cpc = FlatAdam(params, flat, gflat, lr=2e-3, betas=(0.8, 0.99))
        gen = torch.Generator().manual_seed(9)
        for _ in range(3):
            gflat.copy_(torch.randn(gflat.shape, generator=gen).cuda() * 0.01)
            if fused:
                FlatAdam.step_pair(enc, cpc)
            else:
                enc.step()
                cpc.step()
        torch.cuda.synchronize()
        res.append((flat.clone(), enc._m.clone(), enc._v.clone(), cpc._m.clone(), cpc._v.clone(), list(enc._steps),
                    list(cpc._steps)))
    for a, b in zip(res[0][:5], res[1][:5]):
        assert torch.equal(a, b)
    assert res[0][5:] == res[1][5:] == ([3] * 4, [3] * 5)


def test_flat_adam_step_with_target_lerp():
    """FlatAdam.step_with_lerp(critic_optimizer, target, ...) (critic_optimizer.step() and the three soft_update_params
    that follow it, curl_sac.py:367,442-445, in one launch): exactly the parameters, moments and targets of step()
    followed by ops.soft_update2; refused (nothing done) when the optimizer's run is not the announced one."""
    from curla_amd import ops
    from curla_amd.optim import FlatAdam
    sizes = [(32, 9, 3, 3), (32,), (50, 1203), (50,), (64, 52), (64,), (1, 64), (1,)]
    res = []
    for fused in (False, True):
        flat, gflat, params = _flat_params(sizes, "cuda", seed=7)
        n = flat.numel()
        target = (flat + 0.01 * torch.randn(n, generator=torch.Generator().manual_seed(2)).cuda()).contiguous()
        opt = FlatAdam(params, flat, gflat, lr=1e-3, betas=(0.9, 0.999))
        lo, hi = opt._lo, opt._hi
        split = (params[4].data_ptr() - flat.data_ptr()) // 4 - lo  # "encoder" = the first four tensors
        gen = torch.Generator().manual_seed(13)
        for _ in range(3):
            gflat.copy_(torch.randn(gflat.shape, generator=gen).cuda() * 0.01)
            if fused:
                assert not FlatAdam.step_with_lerp(opt, target, lo + 4, hi, split, 0.05, 0.01)  # not its run: refused
                assert FlatAdam.step_with_lerp(opt, target, lo, hi, split, 0.05, 0.01)
            else:
                opt.step()
                ops.soft_update2(flat[lo:hi], target[lo:hi], split, 0.05, 0.01)
        torch.cuda.synchronize()
        res.append((flat.clone(), opt._m.clone(), opt._v.clone(), target.clone(), list(opt._steps)))
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    assert res[0][4] == res[1][4] == [3] * len(sizes)


def test_flat_adam_step_with_float64_scalar():
    """FlatAdam.step_with_scalar(actor_optimizer, log_alpha_optimizer) (curl_sac.py:393-404 in one launch): the flat
    parameters as FlatAdam.step() leaves them, the float64 scalar as torch's own Adam steps it."""
    from curla_amd.optim import FlatAdam
    sizes = [(50, 120), (50,), (64, 50), (64,)]
    gen = torch.Generator().manual_seed(11)
    grads = [(torch.randn(sum(int(np.prod(z)) + 3 & ~3 for z in sizes) + 8, generator=gen) * 0.01,
              torch.randn((), generator=gen, dtype=torch.float64)) for _ in range(4)]
    res = []
    for fused in (False, True):
        flat, gflat, params = _flat_params(sizes, "cuda", seed=5)
        a = torch.tensor(np.log(0.1), device="cuda", requires_grad=True)
        assert a.dtype == torch.float64
        a.grad = torch.zeros((), device="cuda", dtype=torch.float64)
        opt = FlatAdam(params, flat, gflat, lr=1e-3)
        sopt = torch.optim.Adam([a], lr=1e-4, betas=(0.5, 0.999), foreach=False)
        for gf, ga in grads:
            gflat.copy_(gf[:gflat.numel()].cuda())
            a.grad.copy_(ga)
            if fused:
                FlatAdam.step_with_scalar(opt, sopt)
            else:
                opt.step()
                sopt.step()
        torch.cuda.synchronize()
        st = sopt.state[a]
        res.append((flat.clone(), opt._m.clone(), opt._v.clone(), a.detach().clone(), st["exp_avg"].clone(),
                    st["exp_avg_sq"].clone(), float(st["step"]), list(opt._steps)))
    for x, y in zip(res[0][:3], res[1][:3]):
        assert torch.equal(x, y)
    for x, y in zip(res[0][3:6], res[1][3:6]):  # float64: same formula, torch may contract differently
        assert abs(float(x) - float(y)) <= 4e-16 * max(1.0, abs(float(x))), (float(x), float(y))
    assert res[0][6:] == res[1][6:] == (4.0, [4] * 4)
    assert float(res[1][3]) != float(np.log(0.1))


@pytest.mark.parametrize("hw", [(13, 13), (9, 75)])  # (75 wide: rows of >= 16 pixel quads through all three layers)
@pytest.mark.parametrize("two", [False, True])
def test_conv_s1_forward_stack_one_launch(ops, s1_impl, two, hw):
    """curla_conv3x3_s1_fwd_stack: three stride-1 layers of one or two minibatches in ONE launch (a workgroup owns its
    samples through the layers) -- every layer's activations bit-identical to the per-layer launches.  Batch sizes
    must be multiples of the persistent grid (ops.stack_granule()); anything else is refused."""
    G = ops.stack_granule()
    H, W = hw
    B1, B2 = 2 * G, G
    L = 3
    x1, x2 = torch.relu(rnd(B1, H, W, 32, seed=101)).cuda(), torch.relu(rnd(B2, H, W, 32, seed=102)).cuda()
    w1 = [(rnd(32, 32, 3, 3, seed=110 + i) * 0.1).cuda() for i in range(L)]
    w2 = [(rnd(32, 32, 3, 3, seed=120 + i) * 0.1).cuda() for i in range(L)]
    b1 = [(rnd(32, seed=130 + i) * 0.1).cuda() for i in range(L)]
    b2 = [(rnd(32, seed=140 + i) * 0.1).cuda() for i in range(L)]
    mk = lambda B: [torch.full((B, H - 2 * (i + 1), W - 2 * (i + 1), 32), float("nan"), device="cuda") for i in range(L)]  # noqa: E731
    r1, r2, o1, o2 = mk(B1), mk(B2), mk(B1), mk(B2)
    for i in range(L):
        ops.conv_s1_fwd(x1 if i == 0 else r1[i - 1], w1[i], b1[i], r1[i])
        ops.conv_s1_fwd(x2 if i == 0 else r2[i - 1], w2[i], b2[i], r2[i])
    if two:
        assert ops.conv_s1_fwd_stack(x1, w1, b1, o1, x2, w2, b2, o2)
    else:
        assert ops.conv_s1_fwd_stack(x1, w1, b1, o1) and ops.conv_s1_fwd_stack(x2, w2, b2, o2)
    for i in range(L):
        assert torch.equal(o1[i], r1[i]) and torch.equal(o2[i], r2[i]), i
    assert not ops.conv_s1_fwd_stack(x1[:G + 1], w1, b1, [t[:G + 1] for t in o1])


def test_conv_s1_forward_stack_full_size(ops, s1_impl):
    """The stack launch at BASELINE configs[1] size (the critic phase's [obs | next_obs] online pass of 1024 samples
    + the target pass of 512, 37x37 -> 35 -> 33 -> 31): every layer bit-identical to per-layer launches, twice in
    a row into the same buffers (a workgroup reads back its own stores of the previous layer: a stale or late line
    would show up here)."""
    G = ops.stack_granule()
    if 1024 % G or 512 % G:
        pytest.skip("grid size does not divide the BASELINE batch sizes on this device")
    H = W = 37
    L = 3
    g = torch.Generator(device="cuda").manual_seed(7)
    x1 = torch.relu(torch.randn(1024, H, W, 32, device="cuda", generator=g))
    x2 = torch.relu(torch.randn(512, H, W, 32, device="cuda", generator=g))
    w1 = [torch.randn(32, 32, 3, 3, device="cuda", generator=g) * 0.1 for _ in range(L)]
    w2 = [torch.randn(32, 32, 3, 3, device="cuda", generator=g) * 0.1 for _ in range(L)]
    b1 = [torch.randn(32, device="cuda", generator=g) * 0.1 for _ in range(L)]
    b2 = [torch.randn(32, device="cuda", generator=g) * 0.1 for _ in range(L)]
    mk = lambda B: [torch.empty(B, H - 2 * (i + 1), W - 2 * (i + 1), 32, device="cuda") for i in range(L)]  # noqa: E731
    r1, r2, o1, o2 = mk(1024), mk(512), mk(1024), mk(512)
    for i in range(L):
        ops.conv_s1_fwd(x1 if i == 0 else r1[i - 1], w1[i], b1[i], r1[i])
        ops.conv_s1_fwd(x2 if i == 0 else r2[i - 1], w2[i], b2[i], r2[i])
    for rep in range(2):
        for t in o1 + o2:
            t.fill_(float("nan"))
        assert ops.conv_s1_fwd_stack(x1, w1, b1, o1, x2, w2, b2, o2)
        for i in range(L):
            assert torch.equal(o1[i], r1[i]) and torch.equal(o2[i], r2[i]), (rep, i)


def test_f64_rider_kernels_match_the_host_arithmetic_bit_for_bit():
    """curla_f64_pack / curla_f64_unpack against ops.f64_words_of / f64_of_words (tests/test_host_logic.py pins those to
    exact rational arithmetic): the words, and the decoded mean of 2 / 4 / 8 ranks' values after a SUM or an AVG."""
    from curla_amd import ops
    from tests.test_host_logic import _rider_cases
    vals = _rider_cases() + [2.0 ** 30, float("inf")]
    dev = torch.device("cuda")
    v = torch.zeros((), dtype=torch.float64, device=dev)
    words = torch.zeros(ops.F64_WORDS, dtype=torch.float32, device=dev)
    packed = []
    for x in vals:
        v.fill_(x)
        ops.f64_pack(v, words)
        got = words.cpu().tolist()
        assert got == [float(np.float32(t)) for t in ops.f64_words_of(x)], x
        ops.f64_unpack(words, 1, 1, v)
        assert float(v) == x or (np.isinf(x) and np.isinf(float(v))), (x, float(v))
        packed.append(words.clone())
    rs = np.random.RandomState(9)
    n_finite = len(vals) - 2
    for world in (2, 4, 8):
        for _ in range(30):
            pick = rs.randint(0, n_finite, world)
            s = torch.stack([packed[i] for i in pick]).sum(0)
            for w_in, n_mul in ((s, 1), (s / world, world)):
                ops.f64_unpack(w_in.contiguous(), n_mul, world, v)
                assert float(v) == ops.f64_of_words(w_in.cpu().tolist(), n_mul, world), (world, pick)
            if world == 2:
                a, b = np.float64(vals[pick[0]]), np.float64(vals[pick[1]])
                assert float(v) == (a + b) / np.float64(2)


@pytest.mark.parametrize("nf", [16, 64, 8])
def test_generic_filter_counts_against_pytorch(ops, nf):
    """num_filters other than 32 (encoder.py:54-63, train.py:84) run plain direct convolutions behind the same C entry
    points (csrc/conv_generic.h): the first layer from the uint8 ring (gather + crop) and from float NCHW / NHWC tensors,
    a stride-1 layer, both data and weight gradients, the one-launch backward and the multi-layer slab reduction."""
    from oracle import curla_oracle as O
    C, Hs, Ws, Hc, Wc, B, N = 9, 30, 34, 25, 29, 5, 7
    frames, ring = _ring(N, C, Hs, Ws, seed=nf)
    rs = np.random.RandomState(3)
    idx = rs.randint(0, N, B)
    h1, w1 = rs.randint(0, Hs - Hc + 1, B).astype(np.int32), rs.randint(0, Ws - Wc + 1, B).astype(np.int32)
    crop = torch.from_numpy(O.random_crop(frames[idx], h1, w1, (Hc, Wc)).astype(np.float32))
    w0, b0 = rnd(nf, C, 3, 3, seed=11, scale=0.2), rnd(nf, seed=12, scale=0.1)
    w1_, b1_ = rnd(nf, nf, 3, 3, seed=13, scale=0.1), rnd(nf, seed=14, scale=0.1)
    w0r, b0r, w1r, b1r = (t.clone().requires_grad_(True) for t in (w0, b0, w1_, b1_))
    a1 = torch.relu(F.conv2d(crop / 255.0, w0r, b0r, stride=2))
    a2 = torch.relu(F.conv2d(a1, w1r, b1r))
    g2 = rnd(*a2.shape, seed=15) * (a2 > 0)
    a2.backward(g2)
    Ho, Wo = a1.shape[2:]
    # forward: the three first-layer sources, then the stride-1 layer
    obs_u8 = ops.ObsRef.from_ring(ring, dev(idx.astype(np.int64)), dev(h1), dev(w1), B, (Hc, Wc))
    srcs = [("u8 ring", obs_u8), ("f32 NCHW", ops.ObsRef.from_tensor(dev(crop))),
            ("f32 NHWC", ops.ObsRef.from_nhwc(nhwc(crop)))]
    d_w0, d_b0, d_w1, d_b1 = dev(w0), dev(b0), dev(w1_), dev(b1_)
    for name, o in srcs:
        out1 = torch.full((B, Ho, Wo, nf), float("nan"), device="cuda")
        ops.conv1_fwd(o, d_w0, d_b0, out1)
        check(f"generic conv1_fwd [{name}] nf{nf}", nchw(out1), a1.detach())
    out2 = torch.full((B, Ho - 2, Wo - 2, nf), float("nan"), device="cuda")
    ops.conv_s1_fwd(out1, d_w1, d_b1, out2)
    check(f"generic conv_s1_fwd nf{nf}", nchw(out2), a2.detach())
    if nf == 16:  # two problems per call: conv1_fwd2 / conv_s1_fwd2 run them one after the other
        o1b, o2b = torch.full_like(out1, float("nan")), torch.full_like(out2, float("nan"))
        o1c = torch.full_like(out1, float("nan"))
        ops.conv1_fwd2(obs_u8, d_w0, d_b0, o1b, obs_u8, d_w0, d_b0, o1c)
        assert torch.equal(o1b, out1) and torch.equal(o1c, out1)
        o2c = torch.full_like(out2, float("nan"))
        ops.conv_s1_fwd2(out1, d_w1, d_b1, o2b, out1, d_w1, d_b1, o2c)
        assert torch.equal(o2b, out2) and torch.equal(o2c, out2)
        assert not ops.conv_s1_fwd_stack(out1, [d_w1], [d_b1], [o2b])  # (the stack launch is the 32-filter kernels')
    # backward of the stride-1 layer in one call (weight-gradient slab + data gradient), then the first layer's
    # weight gradient from each source, all slabs summed by ONE reduction launch
    g2d = nhwc(g2)
    g1 = torch.full((B, Ho, Wo, nf), float("nan"), device="cuda")
    ws1 = torch.empty(ops.wgrad_workspace_floats(nf), device="cuda")
    n1 = ops.conv_s1_bwd_slabs(out1, g2d, d_w1, g1, ws1)
    a1d = a1.detach().clone().requires_grad_(True)
    torch.relu(F.conv2d(a1d, w1_, b1_)).backward(g2)
    check(f"generic dgrad nf{nf}", nchw(g1), a1d.grad * (a1.detach() > 0))
    gd = torch.full_like(g1, float("nan"))
    ops.conv_s1_dgrad(g2d, d_w1, out1, gd)
    assert torch.equal(gd, g1)
    for name, o in srcs:
        ws0 = torch.empty(ops.wgrad_workspace_floats(C), device="cuda")
        n0 = ops.conv1_wgrad_slabs(o, g1, ws0, nf)
        dw0, db0 = torch.full((nf, C, 3, 3), float("nan"), device="cuda"), torch.full((nf,), float("nan"), device="cuda")
        dw1, db1 = torch.full((nf, nf, 3, 3), float("nan"), device="cuda"), torch.full((nf,), float("nan"), device="cuda")
        ops.wgrad_reduce_multi([(ws1, n1, dw1, db1), (ws0, n0, dw0, db0)])
        check(f"generic wgrad layer 2 dW nf{nf}", dw1.cpu(), w1r.grad)
        check(f"generic wgrad layer 2 db nf{nf}", db1.cpu(), b1r.grad)
        check(f"generic wgrad layer 1 dW [{name}] nf{nf}", dw0.cpu(), w0r.grad)
        check(f"generic wgrad layer 1 db [{name}] nf{nf}", db0.cpu(), b0r.grad)
    # the direct (slab + reduce in one call) entry points
    dwa, dba = torch.full((nf, nf, 3, 3), float("nan"), device="cuda"), torch.full((nf,), float("nan"), device="cuda")
    ops.conv_s1_wgrad(out1, g2d, dwa, dba, ws1)
    check(f"generic conv_s1_wgrad dW nf{nf}", dwa.cpu(), w1r.grad)
    dwb, dbb = torch.full((nf, C, 3, 3), float("nan"), device="cuda"), torch.full((nf,), float("nan"), device="cuda")
    ops.conv1_wgrad(obs_u8, g1, dwb, dbb, ws0)
    check(f"generic conv1_wgrad dW nf{nf}", dwb.cpu(), w0r.grad)
    check(f"generic conv1_wgrad db nf{nf}", dbb.cpu(), b0r.grad)
