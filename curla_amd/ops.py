"""Tensor-level wrappers over the C ABI (include/curla_hip.h).

Every function enqueues HIP kernels on torch's current stream and returns
immediately; tensors are only used for their device pointers.  Nothing here
computes on the host and nothing falls back to PyTorch ops.
"""
import ctypes

import torch

from . import _lib
from ._lib import call, ptr, stream

F32 = torch.float32


def _dev(t, dtype=F32):
    if not ((t.is_cuda or _lib._trace_hook is not None) and t.dtype == dtype and t.is_contiguous()):
        raise _lib.CurlaHipError(f"expected a contiguous {dtype} CUDA tensor, got {t.dtype} {t.device} "
                                 f"contiguous={t.is_contiguous()}")
    return t


class ObsRef:
    """Where a minibatch's pixels live: the uint8 NHWC replay ring plus per-sample
    frame indices and crop offsets (is_u8 = 1: fused gather+crop, the fast path),
    a float NCHW tensor in [0,255] (is_u8 = 0: the reference's tensor contract) or
    a float NHWC tensor (is_u8 = 2: output of the float augmentations)."""

    __slots__ = ("src", "is_u8", "idx", "h1", "w1", "B", "C", "Hs", "Ws", "Hc", "Wc", "guard", "pair")

    def __init__(self):
        self.guard = None
        # (merged handle, second handle): this minibatch followed by ``second``'s as ONE handle of 2B observations
        # (set by ReplayBuffer.sample_cpc_refs on obs for (obs | next_obs))
        self.pair = None

    @staticmethod
    def from_ring(frames, idx, h1, w1, B, crop_hw, guard=None):
        """``guard`` = (generation list, slot, generation): the replay buffer's device index block (and assembled
        stacks) this handle points into is recycled after a few further samples; check() raises once it has been."""
        o = ObsRef()
        _dev(frames, torch.uint8)
        o.src, o.is_u8, o.idx, o.h1, o.w1, o.B = frames, 1, idx, h1, w1, B
        _, o.Hs, o.Ws, o.C = frames.shape
        o.Hc, o.Wc = crop_hw
        o.guard = guard
        return o

    def check(self):
        g = self.guard
        if g is not None and g[0][g[1]] != g[2]:
            raise _lib.CurlaHipError("stale minibatch handle: the replay buffer has drawn further samples since and "
                                     "recycled this one's device index block (ReplayBuffer.N_SAMPLE_SLOTS handles "
                                     "are valid at a time)")
        return self

    @staticmethod
    def from_nhwc(x):
        """float32 NHWC minibatch [B, H, W, C] in [0,255] (augmented observations)."""
        o = ObsRef()
        _dev(x)
        o.src, o.is_u8, o.idx, o.h1, o.w1 = x, 2, None, None, None
        o.B, o.Hc, o.Wc, o.C = x.shape
        o.Hs, o.Ws = o.Hc, o.Wc
        return o

    @staticmethod
    def from_tensor(x):
        o = ObsRef()
        _dev(x)
        o.src, o.is_u8, o.idx, o.h1, o.w1 = x, 0, None, None, None
        o.B, o.C, o.Hc, o.Wc = x.shape
        o.Hs, o.Ws = o.Hc, o.Wc
        return o


def conv1_fwd(obs: ObsRef, w, b, out, scale=1.0 / 255.0):
    obs.check()
    call("curla_conv1_fwd", ptr(obs.src), obs.is_u8, ptr(obs.idx), ptr(obs.h1), ptr(obs.w1), ptr(w), ptr(b), ptr(out),
         obs.B, obs.C, obs.Hs, obs.Ws, obs.Hc, obs.Wc, w.shape[0], scale, stream())


def conv1_wgrad(obs: ObsRef, g, dw, db, ws, scale=1.0 / 255.0):
    obs.check()
    call("curla_conv1_wgrad", ptr(obs.src), obs.is_u8, ptr(obs.idx), ptr(obs.h1), ptr(obs.w1), ptr(g), ptr(dw), ptr(db),
         ptr(ws), obs.B, obs.C, obs.Hs, obs.Ws, obs.Hc, obs.Wc, dw.shape[0], scale, stream())


def conv_s1_fwd(x, w, b, out):
    B, H, W, C = x.shape
    call("curla_conv3x3_s1_fwd", ptr(x), ptr(w), ptr(b), ptr(out), B, H, W, C, stream())


def conv_s1_fwd2(x, w, b, out, x2, w2, b2, out2):
    """Two forwards of the same geometry (own weights each) in one launch."""
    B, H, W, C = x.shape
    assert x2.shape[1:] == x.shape[1:]
    call("curla_conv3x3_s1_fwd2", ptr(x), ptr(w), ptr(b), ptr(out), B, ptr(x2), ptr(w2), ptr(b2), ptr(out2), x2.shape[0],
         H, W, C, stream())


def conv_s1_fwd_stack(x, ws, bs, outs, x2=None, ws2=None, bs2=None, outs2=None):
    """All stride-1 layers of one or two minibatches in ONE launch (x -> outs[0] -> outs[1] ...); returns False when
    the kernels cannot (batch sizes not multiples of the persistent grid: use one conv_s1_fwd2 per layer)."""
    import ctypes
    B, H, W, C = x.shape
    n = len(ws)
    B2 = 0 if x2 is None else x2.shape[0]
    G = stack_granule()
    if n > 6 or B % G or B2 % G or C != 32:  # (other filter counts: the generic per-layer kernels, csrc/conv_generic.h)
        return False
    P = ctypes.c_void_p * n
    arr = lambda ts: P(*[ptr(t) for t in ts])  # noqa: E731
    a_w, a_b, a_o = arr(ws), arr(bs), arr(outs)
    if B2:
        assert x2.shape[1:] == x.shape[1:]
        a_w2, a_b2, a_o2 = arr(ws2), arr(bs2), arr(outs2)
        p2 = (ptr(x2), ctypes.addressof(a_w2), ctypes.addressof(a_b2), ctypes.addressof(a_o2))
    else:
        p2 = (None, None, None, None)
    call("curla_conv3x3_s1_fwd_stack", n, ptr(x), ctypes.addressof(a_w), ctypes.addressof(a_b), ctypes.addressof(a_o), B,
         p2[0], p2[1], p2[2], p2[3], B2, H, W, C, stream())
    return True


_PER_DEVICE = {}  # (what, device index) -> value: a process may drive several GPUs


def _device_cached(what, compute):
    key = (what, torch.cuda.current_device())
    if key not in _PER_DEVICE:
        _PER_DEVICE[key] = compute()
    return _PER_DEVICE[key]


def stack_granule():
    """Batch-size multiple conv_s1_fwd_stack needs (its persistent grid: one workgroup per CU, two in the banded form)."""
    if _lib._trace_hook is not None:
        return 256  # (launch-schedule tests without a device: MI355X's CU count)
    return _device_cached("stack_granule", lambda: int(_lib.load().curla_conv3x3_s1_stack_granule()))


def cu_count():
    if _lib._trace_hook is not None:
        return 256  # (launch-schedule tests without a device: MI355X's count)
    return _device_cached("cus", lambda: torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)


def conv1_pairable(o1: "ObsRef", o2: "ObsRef"):
    """Can the first-layer forwards of two handles share a launch?  Both uint8, from the same ring, same crop."""
    return (o1.is_u8 == 1 and o2.is_u8 == 1 and o1.src.data_ptr() == o2.src.data_ptr() and o1.src.shape == o2.src.shape
            and (o1.Hc, o1.Wc) == (o2.Hc, o2.Wc))


def conv1_fwd2(o1: "ObsRef", w, b, out, o2: "ObsRef", w2, b2, out2, scale=1.0 / 255.0):
    o1.check(), o2.check()
    call("curla_conv1_fwd2", ptr(o1.src), ptr(o1.idx), ptr(o1.h1), ptr(o1.w1), ptr(w), ptr(b), ptr(out), o1.B, ptr(o2.idx),
         ptr(o2.h1), ptr(o2.w1), ptr(w2), ptr(b2), ptr(out2), o2.B, o1.C, o1.Hs, o1.Ws, o1.Hc, o1.Wc, w.shape[0], scale,
         stream())


def conv_s1_dgrad(g, w, act_below, gin):
    B, H, W, C = g.shape
    call("curla_conv3x3_s1_dgrad", ptr(g), ptr(w), ptr(act_below), ptr(gin), B, H, W, C, stream())


def conv_s1_wgrad(x, g, dw, db, ws):
    B, H, W, C = x.shape
    call("curla_conv3x3_s1_wgrad", ptr(x), ptr(g), ptr(dw), ptr(db), ptr(ws), B, H, W, C, stream())


def conv_s1_wgrad_slabs(x, g, ws):
    """The weight-gradient kernel only: per-workgroup partial sums into ``ws``; returns the slab count."""
    import ctypes
    B, H, W, C = x.shape
    n = ctypes.c_int(1)
    call("curla_conv3x3_s1_wgrad_slabs", ptr(x), ptr(g), ptr(ws), B, H, W, C, ctypes.addressof(n), stream())
    return n.value


def conv_s1_bwd_slabs(x, g, w, gin, ws):
    """Weight-gradient slabs (into ``ws``) and the data gradient ``gin`` (masked by x > 0) of one stride-1 layer with
    input ``x`` and output gradient ``g``, in one launch; returns the slab count."""
    import ctypes
    B, H, W, C = x.shape
    n = ctypes.c_int(1)
    call("curla_conv3x3_s1_bwd_slabs", ptr(x), ptr(g), ptr(w), ptr(gin), ptr(ws), B, H, W, C, ctypes.addressof(n),
         stream())
    return n.value


def conv1_wgrad_slabs(obs: ObsRef, g, ws, channels, scale=1.0 / 255.0):
    import ctypes
    obs.check()
    n = ctypes.c_int(1)
    call("curla_conv1_wgrad_slabs", ptr(obs.src), obs.is_u8, ptr(obs.idx), ptr(obs.h1), ptr(obs.w1), ptr(g), ptr(ws),
         obs.B, obs.C, obs.Hs, obs.Ws, obs.Hc, obs.Wc, channels, scale, ctypes.addressof(n), stream())
    return n.value


def wgrad_reduce_multi(jobs):
    """jobs: [(slabs workspace, slab count, dw, db), ...] (<= 8): every layer's slabs summed into dW / db in ONE launch."""
    import ctypes
    n = len(jobs)
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    slabs = P(*[ptr(j[0]) for j in jobs])
    ns = I(*[int(j[1]) for j in jobs])
    nw = I(*[int(j[2].numel()) for j in jobs])
    nb = I(*[int(j[3].numel()) for j in jobs])  # (bias sums behind the weight sums of a slab: the layer's filter count)
    dw = P(*[ptr(j[2]) for j in jobs])
    db = P(*[ptr(j[3]) for j in jobs])
    call("curla_wgrad_reduce_multi", n, ctypes.addressof(slabs), ctypes.addressof(ns), ctypes.addressof(nw),
         ctypes.addressof(nb), ctypes.addressof(dw), ctypes.addressof(db), stream())


def wgrad_workspace_floats(cin):
    if _lib._trace_hook is not None:
        return 512 * (32 * cin * 9 + 32)
    return int(_lib.load().curla_conv_wgrad_workspace_floats(cin))


def gemm(A, a_kmajor, lda, sA, Bm, b_kmajor, ldb, sB, C, ldc, sC, M, N, K, nbatch=1, ksplit=1, split_stride=0,
         alpha=1.0, bias=None, sBias=0, relu=0, mask=None, ldmask=0, sMask=0):
    call("curla_gemm", ptr(A), a_kmajor, lda, sA, ptr(Bm), b_kmajor, ldb, sB, ptr(C), ldc, sC, M, N, K, nbatch, ksplit,
         split_stride, alpha, ptr(bias), sBias, relu, ptr(mask), ldmask, sMask, stream())


def gemm_multi(As, Bs, Cs, M, N, K, ksplit=1, split_stride=0):
    """Up to 4 products C_i = A_i [M,K] @ B_i [N,K]^T of one shape (split-K partials at C_i + s * split_stride) in one
    launch."""
    import ctypes
    n = len(As)
    P = ctypes.c_void_p * n
    a, b, c = P(*[ptr(t) for t in As]), P(*[ptr(t) for t in Bs]), P(*[ptr(t) for t in Cs])
    call("curla_gemm_multi", n, ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(c), K, K, N, M, N, K, ksplit,
         split_stride, stream())


FC_FWD_MAX_SPLIT = 64


def fc_fwd_supported(B, F_, K):
    """Shapes the streaming encoder-fc forward (curla_fc_fwd_multi) is instantiated for."""
    return 50 <= F_ <= 64 and F_ % 2 == 0 and B % 128 == 0 and K % 32 == 0 and B * K * 4 < 2 ** 31 and F_ * K * 4 < 2 ** 31


def fc_fwd_nsplit(nprob, B, K):
    """Split-K factor of the streaming fc forward: two 256-thread workgroups fit a CU, and a grid a few workgroups LARGER
    than that capacity costs a whole extra round (516 workgroups: 72 us, 768: 53 us, 504: 52 us for three [512 x 30752]
    operands) -- so: 1.5 rounds' worth, never just above a whole number of rounds."""
    cap = 2 * cu_count()
    per = nprob * (B // 128)
    ns = max(1, min(FC_FWD_MAX_SPLIT, (3 * cap // 2) // per, K // 32))
    while ns > 1 and ns * per > cap and 0 < (ns * per) % cap <= cap // 8:
        ns -= 1
    return ns


def fc_fwd_multi(xs, Ws, outs, B, F_, K, nsplit, split_stride, blocked=False):
    """Split-K partial sums outs[i][s] = xs[i] [B, K] @ Ws[i] [F, K]^T over split s, up to 4 pairs of one shape in one
    launch (fc_fwd_supported shapes only).  ``blocked``: xs are in the [B/16][K/32][16][32] layout."""
    import ctypes
    n = len(xs)
    P = ctypes.c_void_p * n
    a, b, c = P(*[ptr(t) for t in xs]), P(*[ptr(t) for t in Ws]), P(*[ptr(t) for t in outs])
    call("curla_fc_fwd_multi", n, ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(c), B, F_, K, nsplit,
         split_stride, 1 if blocked else 0, stream())


def to_blocked(x, B, K):
    """Row-major [B, K] -> the blocked layout [B/16][K/32][16][32] (same number of elements, a new tensor)."""
    return x.reshape(B // 16, 16, K // 32, 32).permute(0, 2, 1, 3).contiguous().view(B, K)


def from_blocked(x, B, K):
    return x.reshape(B // 16, K // 32, 16, 32).permute(0, 2, 1, 3).contiguous().view(B, K)


def linear_fwd(x, sx, W, sW, bias, sb, out, so, M, N, K, nb=1, relu=0, outer=None):
    """out[z] = act(x[z] @ W[z]^T + bias[z]);  x [M,K], W [N,K] (nn.Linear layout).
    ``outer`` = (n2, sx2, sW2, so2): a second batch level (item (o, z) at base + z * s + o * s2; bias strides as W's)."""
    if outer is None:
        gemm(x, 0, K, sx, W, 0, K, sW, out, N, so, M, N, K, nb, bias=bias, sBias=sb, relu=relu)
    else:
        n2, sx2, sW2, so2 = outer
        call("curla_gemm_nested", ptr(x), 0, K, sx, sx2, ptr(W), 0, K, sW, sW2, ptr(out), N, so, so2, M, N, K, nb, n2,
             1.0, ptr(bias), sb, sW2, relu, None, 0, 0, 0, stream())


def linear_dx(dy, sdy, W, sW, out, so, M, N, K, nb=1, mask=None, smask=0):
    """out[z] = (dy[z] @ W[z]) masked;  dy [M,N], W [N,K] -> out [M,K]."""
    gemm(dy, 0, N, sdy, W, 1, K, sW, out, K, so, M, K, N, nb, mask=mask, ldmask=K, sMask=smask)


def linear_dw(dy, sdy, x, sx, out, so, M, N, K, nb=1, colsum=None, s_colsum=0):
    """out[z] = dy[z]^T @ x[z];  dy [M,N], x [M,K] -> out [N,K].
    ``colsum`` [N] (batch stride s_colsum; only where linear_dw_folds_bias(...)): also the column sums of dy, the
    bias gradient, from the same launch."""
    if colsum is None:
        gemm(dy, 1, N, sdy, x, 1, K, sx, out, K, so, N, K, M, nb)
    else:
        call("curla_gemm_colsum", ptr(dy), 1, N, sdy, ptr(x), 1, K, sx, ptr(out), K, so, N, K, M, nb, ptr(colsum),
             s_colsum, stream())


def linear_bwd(dy, sdy, x, sx, W, sW, dW, sdW, dx, sdx, M, N, K, nb=1, mask=None, smask=0, db=None, sdb=0):
    """linear_dw + linear_dx of one layer over the same dy [M,N] in ONE launch where both products take the same kernel
    family (two launches otherwise): dW [N,K] = dy^T x, dx [M,K] = (dy W) masked; ``db`` [N] as linear_dw's colsum."""
    call("curla_linear_bwd", ptr(dy), sdy, ptr(x), sx, ptr(W), sW, ptr(mask), smask, ptr(dW), sdW, ptr(db), sdb, ptr(dx),
         sdx, M, N, K, nb, stream())


_small_shape_cache = {}


def linear_dw_folds_bias(M, N, K, nb=1):
    """True when linear_dw(dy [M,N], x [M,K]) can carry the bias gradient (curla_gemm_colsum's conditions)."""
    key = (N, K, M, nb)
    if key not in _small_shape_cache:
        _small_shape_cache[key] = bool(_lib.load().curla_gemm_small_shape(N, K, M, nb))
    return _small_shape_cache[key]


def fc_bwd_streams(F_, K):
    """True when the encoder-fc backward products take the streaming kernels (curla_fc_dx / curla_fc_dw)."""
    return F_ <= 64 and K % 4 == 0


def fc_dx(dz, W, out, B, F_, K, mask=None):
    """out = (dz @ W) masked by ``mask`` > 0;  dz [B,F], W [F,K] (the encoder fc weight), out/mask [B,K]."""
    call("curla_fc_dx", ptr(dz), ptr(W), ptr(mask), ptr(out), B, F_, K, stream())


def fc_dw(dz, x, out, B, F_, K, ln=None):
    """out = dz^T @ x;  dz [B,F], x [B,K] -> out [F,K] (the encoder fc weight gradient).
    ``ln``: the token of a deferred ``ln_bwd`` -- its parameter gradients are finished in this launch."""
    if ln is None:
        call("curla_fc_dw", ptr(dz), ptr(x), ptr(out), B, F_, K, stream())
    else:
        part, n, dgamma, dbeta, dbias = ln
        call("curla_fc_dw_ln", ptr(dz), ptr(x), ptr(out), B, F_, K, ptr(part), n, ptr(dgamma), ptr(dbeta), ptr(dbias),
             stream())


def fc_bwd(dz, W, x, dx, dW, B, F_, K, ln=None):
    """dx = (dz @ W) masked by x > 0 and dW = dz^T @ x in one launch (x = the fc layer's input, a ReLU output).
    ``ln``: as for fc_dw."""
    if ln is None:
        call("curla_fc_bwd", ptr(dz), ptr(W), ptr(x), ptr(dx), ptr(dW), B, F_, K, stream())
    elif len(ln) > 5:  # (curl_head's token: a second set of partial sums rides along)
        part, n, dgamma, dbeta, dbias, xpart, xn, xlen, xout = ln
        call("curla_fc_bwd_ln2", ptr(dz), ptr(W), ptr(x), ptr(dx), ptr(dW), B, F_, K, ptr(part), n, ptr(dgamma),
             ptr(dbeta), ptr(dbias), ptr(xpart), xn, xlen, ptr(xout), stream())
    else:
        part, n, dgamma, dbeta, dbias = ln
        call("curla_fc_bwd_ln", ptr(dz), ptr(W), ptr(x), ptr(dx), ptr(dW), B, F_, K, ptr(part), n, ptr(dgamma),
             ptr(dbeta), ptr(dbias), stream())


def mlp_out_fwd(h, sh, W, sW, bias, sb, out, so, M, N, K, nb=1, outer=None):
    """out[z] = h[z] @ W[z]^T + bias[z] for a last layer with N <= 16 outputs (one wave per row, no GEMM).
    ``outer`` = (n2, sh2, sW2, so2) as in linear_fwd."""
    if outer is None:
        call("curla_mlp_out_fwd", ptr(h), sh, ptr(W), sW, ptr(bias), sb, ptr(out), so, M, N, K, nb, stream())
    else:
        n2, sh2, sW2, so2 = outer
        call("curla_mlp_out_fwd_nested", ptr(h), sh, sh2, ptr(W), sW, sW2, ptr(bias), sb, sW2, ptr(out), so, so2, M, N,
             K, nb, n2, stream())


def mlp_out_bwd(dy, sdy, h, sh, W, sW, dh, sdh, dW, sdW, M, N, K, nb=1, db_out=None, db_hidden=None, sdb=0):
    """dh[z] = (dy[z] @ W[z]) masked by h[z] > 0 and (dW not None) dW[z] = dy[z]^T @ h[z], in one pass over h.
    ``db_out`` [N] / ``db_hidden`` [K] (batch stride sdb): also the column sums of dy / of dh -- the bias gradients of
    this layer and of the one below."""
    if db_out is None and db_hidden is None:
        call("curla_mlp_out_bwd", ptr(dy), sdy, ptr(h), sh, ptr(W), sW, ptr(dh), sdh, ptr(dW), sdW, M, N, K, nb, stream())
    else:
        call("curla_mlp_out_bwd_bias", ptr(dy), sdy, ptr(h), sh, ptr(W), sW, ptr(dh), sdh, ptr(dW), sdW, M, N, K, nb,
             ptr(db_out), ptr(db_hidden), sdb, stream())


class _LossArgs(ctypes.Structure):  # CurlaLossArgs
    _fields_ = [("kind", ctypes.c_int), ("A", ctypes.c_int), ("twin_stride", ctypes.c_longlong)] + \
               [(n, ctypes.c_void_p) for n in ("q", "target_q_twin", "log_pi", "reward", "not_done", "log_std", "log_alpha",
                                               "dlog_alpha", "target_q", "scalars", "dq")] + \
               [("discount", ctypes.c_float), ("target_entropy", ctypes.c_float)]


def mlp_out_bwd_loss(loss, h, sh, W, sW, dh, sdh, dW, sdW, B, K, db_out=None, db_hidden=None, sdb=0):
    """mlp_out_bwd of the twin Q functions' last layer with its output gradient computed from the loss inputs instead
    of read (``loss``: dict of CurlaLossArgs fields; kind 1 = critic TD loss, 2 = actor / alpha loss): the loss
    kernel's launch goes away, its scalars and dq are written by this one."""
    a = _LossArgs()
    a.kind, a.A, a.twin_stride = loss["kind"], loss.get("A", 0), loss["twin_stride"]
    for n in ("q", "target_q_twin", "log_pi", "reward", "not_done", "log_std", "log_alpha", "dlog_alpha", "target_q",
              "scalars", "dq"):
        setattr(a, n, ptr(loss.get(n)))
    a.discount, a.target_entropy = loss.get("discount", 0.0), loss.get("target_entropy", 0.0)
    call("curla_mlp_out_bwd_loss", ctypes.addressof(a), ptr(h), sh, ptr(W), sW, ptr(dh), sdh, ptr(dW), sdW, B, K,
         ptr(db_out), ptr(db_hidden), sdb, stream())


MLP_OUT_MAX = 16


def fc_ln_fwd(partial, nsplit, split_stride, ldp, bias, gamma, beta, B, F, y, fc_out=None, xhat=None, rstd=None,
              eps=1e-5, tanh_out=0, xa=None, act=None):
    """``xa`` [B, F + A] with ``act`` [B, A]: also writes the rows [y | act] (the Q functions' input)."""
    A = 0 if xa is None else xa.shape[1] - F
    call("curla_fc_ln_fwd", ptr(partial), nsplit, split_stride, ldp, ptr(bias), ptr(gamma), ptr(beta), B, F, eps,
         ptr(fc_out), ptr(y), ptr(xhat), ptr(rstd), tanh_out, ptr(xa), ptr(act), A, stream())


_FC_LN_PTRS = ("partial", "bias", "gamma", "beta", "fc_out", "y", "xhat", "rstd", "xa", "act")


class _FcLnJob(ctypes.Structure):  # CurlaFcLnJob
    _fields_ = [(n, ctypes.c_void_p) for n in _FC_LN_PTRS] + [("tanh_out", ctypes.c_int)]


def fc_ln_fwd_multi(jobs, nsplit, split_stride, ldp, B, F, eps, A):
    """``jobs``: up to 4 dicts with fc_ln_fwd's per-problem tensors (partial, bias, gamma, beta, y and optionally
    fc_out, xhat, rstd, xa, act, tanh_out) -- one launch."""
    arr = (_FcLnJob * len(jobs))()
    for a, j in zip(arr, jobs):
        for n in _FC_LN_PTRS:
            setattr(a, n, ptr(j.get(n)))
        a.tanh_out = int(j.get("tanh_out", 0))
    call("curla_fc_ln_fwd_multi", len(jobs), ctypes.addressof(arr), nsplit, split_stride, ldp, B, F, eps, A, stream())


def ln_partial_floats(B, F):
    """Size of the partial-sum buffer ``ln_bwd(..., defer=)`` fills: [ceil(B / 4)][3][F]."""
    return ((B + 3) // 4) * 3 * F


def ln_bwd(dy, xhat, rstd, gamma, B, F, dx, dgamma=None, dbeta=None, dbias_in=None, dy2=None, ld=None, defer=None):
    """LayerNorm backward; ``dbias_in`` (optional) receives the column sums of dx (the fc bias gradient).
    The incoming gradient may be ``dy[:, :F] + dy2[:, :F]`` of two row blocks with row stride ``ld`` (the twin halves
    of d(loss)/d[z | a], read in place).
    ``defer`` = a float buffer of ln_partial_floats(B, F): the parameter gradients are left as partial sums in it and
    the returned token must be handed to the ``fc_bwd`` / ``fc_dw`` that follows (``ln=``), which finishes them in its
    own launch.  Returns None when nothing was deferred."""
    if defer is not None and dgamma is not None:
        import ctypes
        n = ctypes.c_int(0)
        call("curla_ln_bwd_partial", ptr(dy), ptr(dy2), F if ld is None else ld, ptr(xhat), ptr(rstd), ptr(gamma), B, F,
             ptr(dx), ptr(defer), ctypes.addressof(n), stream())
        return (defer, n.value, dgamma, dbeta, dbias_in)
    if dy2 is None and ld in (None, F):
        call("curla_ln_bwd", ptr(dy), ptr(xhat), ptr(rstd), ptr(gamma), B, F, ptr(dx), ptr(dgamma), ptr(dbeta),
             ptr(dbias_in), stream())
    else:
        call("curla_ln_bwd_twin", ptr(dy), ptr(dy2), F if ld is None else ld, ptr(xhat), ptr(rstd), ptr(gamma), B, F,
             ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dbias_in), stream())


def colsum(X, M, N, ldx, sX, out, sOut, nb=1):
    call("curla_colsum", ptr(X), M, N, ldx, sX, ptr(out), sOut, nb, stream())


def colsum3(X0, N0, X1, N1, X2, N2, M, out0, out1, out2, sOut, nb=1):
    """Three column sums (dense [nb][M][N_i] inputs) in one launch."""
    call("curla_colsum3", ptr(X0), N0, ptr(X1), N1, ptr(X2), N2, M, ptr(out0), ptr(out1), ptr(out2), sOut, nb, stream())


def actor_head_fwd(trunk_out, noise, B, A, lo, hi, mu=None, pi=None, log_pi=None, log_std=None, tanh_ls=None, xa=None,
                   rng=None):
    """``xa`` [B, F + A]: pi is also written into its last A columns (the Q functions' input rows).
    ``rng`` = (seed, offset[, device address]): the noise is drawn inside the launch (Philox stream, see curla_hip.h) and
    WRITTEN to ``noise``; the caller owns the offset bookkeeping (ceil(B A / 4) counters per call).  With a device
    address the kernel reads (seed, offset) from there when it runs (captured update graphs)."""
    pi_xa, ld = (None, 0) if xa is None else (xa.data_ptr() + 4 * (xa.shape[1] - A), xa.shape[1])
    if rng is not None:
        call("curla_actor_head_fwd_rng", ptr(trunk_out), ptr(noise), int(rng[0]) & (2 ** 64 - 1), int(rng[1]),
             rng[2] if len(rng) > 2 else None, B, A, lo, hi, ptr(mu), ptr(pi), ptr(log_pi), ptr(log_std), ptr(tanh_ls),
             pi_xa, ld, stream())
        return
    call("curla_actor_head_fwd", ptr(trunk_out), ptr(noise), B, A, lo, hi, ptr(mu), ptr(pi), ptr(log_pi), ptr(log_std),
         ptr(tanh_ls), pi_xa, ld, stream())


def mlp_out_head_fwd(h, W, bias, trunk_out, noise, B, A, K, lo, hi, mu=None, pi=None, log_pi=None, log_std=None,
                     tanh_ls=None, xa=None, rng=None):
    """The actor trunk's last layer (h [B, K] -> trunk_out [B, 2A]) with the policy head (actor_head_fwd) run by the
    same launch.  ``rng``: as for actor_head_fwd."""
    pi_xa, ld = (None, 0) if xa is None else (xa.data_ptr() + 4 * (xa.shape[1] - A), xa.shape[1])
    if rng is not None:
        call("curla_mlp_out_head_fwd_rng", ptr(h), ptr(W), ptr(bias), ptr(trunk_out), B, A, K, ptr(noise),
             int(rng[0]) & (2 ** 64 - 1), int(rng[1]), rng[2] if len(rng) > 2 else None, lo, hi, ptr(mu), ptr(pi),
             ptr(log_pi), ptr(log_std), ptr(tanh_ls), pi_xa, ld, stream())
        return
    call("curla_mlp_out_head_fwd", ptr(h), ptr(W), ptr(bias), ptr(trunk_out), B, A, K, ptr(noise), lo, hi, ptr(mu),
         ptr(pi), ptr(log_pi), ptr(log_std), ptr(tanh_ls), pi_xa, ld, stream())


def actor_head_bwd(gpi, log_alpha, glp_scale, noise, pi, log_std, tanh_ls, B, A, lo, hi, dtrunk_out, glp_rows=None,
                   twin_dxa=None, F=0):
    """``twin_dxa`` [2, B, F + A] (instead of ``gpi``): d(loss)/d(action) is read in place as the sum over the twin
    of the action columns of the Q-input gradient (torch.cat's backward, curl_sac.py:138)."""
    if twin_dxa is not None:
        base = ptr(twin_dxa) + 4 * F
        g1, g2, ld = base, base + 4 * B * (F + A), F + A
    else:
        g1, g2, ld = ptr(gpi), None, A
    call("curla_actor_head_bwd", g1, g2, ld, ptr(glp_rows), ptr(log_alpha), glp_scale, ptr(noise), ptr(pi),
         ptr(log_std), ptr(tanh_ls), B, A, lo, hi, ptr(dtrunk_out), stream())


def concat(z, act, B, F, A, xa):
    call("curla_concat", ptr(z), ptr(act), B, F, A, ptr(xa), stream())


def split_sum(dxa, twin_stride, B, F, A, dz=None, dact=None):
    call("curla_split_sum", ptr(dxa), twin_stride, B, F, A, ptr(dz), ptr(dact), stream())


def td_target(tq, twin_stride, log_pi, reward, not_done, log_alpha, discount, B, target_q):
    call("curla_td_target", ptr(tq), twin_stride, ptr(log_pi), ptr(reward), ptr(not_done), ptr(log_alpha), discount, B,
         ptr(target_q), stream())


def critic_loss(q, twin_stride, target_q, B, loss, dq):
    call("curla_critic_loss", ptr(q), twin_stride, ptr(target_q), B, ptr(loss), ptr(dq), stream())


def critic_td_loss(q, tq, twin_stride, log_pi, reward, not_done, log_alpha, discount, B, target_q, loss, dq):
    call("curla_critic_td_loss", ptr(q), ptr(tq), twin_stride, ptr(log_pi), ptr(reward), ptr(not_done), ptr(log_alpha),
         discount, B, ptr(target_q), ptr(loss), ptr(dq), stream())


def actor_loss(q, twin_stride, log_pi, log_std, A, log_alpha, target_entropy, B, scalars4, dq, dlog_alpha):
    call("curla_actor_loss", ptr(q), twin_stride, ptr(log_pi), ptr(log_std), A, ptr(log_alpha), target_entropy, B,
         ptr(scalars4), ptr(dq), ptr(dlog_alpha), stream())


def curl_head_supported(B, F_):
    return B % 128 == 0 and B <= 1024 and 49 <= F_ <= 52


def curl_head(z_a, z_pos, wz, xhat, rstd, gamma, B, F_, row_loss, dfc, ln_partial, w_partial, dgamma, dbeta, dbias, dW,
              loss=None, logits=None, dlogits=None, dz=None):
    """The CURL head in one launch (curla_curl_head).  Returns the token for ``fc_bwd(..., ln=)``, which finishes
    dgamma / dbeta / dbias (LayerNorm and fc-bias gradients of the anchor encoder) and dW (CURL.W's gradient)."""
    import ctypes
    n = ctypes.c_int(0)
    call("curla_curl_head", ptr(z_a), ptr(z_pos), ptr(wz), ptr(xhat), ptr(rstd), ptr(gamma), B, F_, ptr(row_loss),
         ptr(loss), ptr(dfc), ptr(ln_partial), ctypes.addressof(n), ptr(w_partial), ptr(logits), ptr(dlogits), ptr(dz),
         stream())
    nparts = n.value if n.value else B // 16  # (call tracing without a device leaves the count untouched)
    return (ln_partial, nparts, dgamma, dbeta, dbias, w_partial, nparts, F_ * F_, dW)


def curl_ce(logits, B, ld, row_loss, loss, dlogits=None):
    call("curla_curl_ce", ptr(logits), B, ld, ptr(row_loss), ptr(loss), ptr(dlogits), stream())


def mean(x, n, out):
    call("curla_mean", ptr(x), n, ptr(out), stream())


# ---- a float64 scalar inside a float32 all-reduce bucket (curla_hip.h: curla_f64_pack / curla_f64_unpack) ------------
F64_WORDS = 8     # CURLA_F64_WORDS
_F64_TOP_EXP = 8  # digit j is worth 2^(8 - 20 j)


def f64_words_of(value):
    """The words curla_f64_pack writes for ``value``, on the host (exact: Python floats are IEEE doubles)."""
    import math
    r = float(value)
    if not abs(r) < 2.0 ** 28:
        return [float(r * 2.0 ** -8)] + [0.0] * (F64_WORDS - 1)
    out = []
    for j in range(F64_WORDS):
        d = float(math.trunc(math.ldexp(r, 20 * j - _F64_TOP_EXP)))
        r -= math.ldexp(d, _F64_TOP_EXP - 20 * j)
        out.append(d)
    return out


def f64_of_words(words, n_mul=1.0, n_div=1.0):
    """What curla_f64_unpack computes, on the host: the exact sum of the digits (Python integers), rounded once."""
    import math
    xs = [float(w) * float(n_mul) for w in words]
    if not all(abs(x) < 2.0 ** 40 for x in xs):  # (nan compares false)
        return sum(math.ldexp(x, _F64_TOP_EXP - 20 * j) for j, x in reversed(list(enumerate(xs)))) / float(n_div)
    total = 0
    for x in xs:
        total = (total << 20) + int(round(x))  # (round(): half to even, as llrint)
    # int -> float is correctly rounded (half to even); the power-of-two scale is exact
    return math.ldexp(float(total), _F64_TOP_EXP - 20 * (F64_WORDS - 1)) / float(n_div)


def f64_pack(value, words):
    """``value``: a one-element float64 tensor; ``words``: F64_WORDS float32 elements (same device).  Host tensors (the
    gloo path of the CPU tests) take the host arithmetic above -- the same digits, bit for bit."""
    assert value.dtype == torch.float64 and value.numel() == 1 and words.dtype == F32 and words.numel() == F64_WORDS
    if not words.is_cuda and _lib._trace_hook is None:
        words.copy_(torch.tensor(f64_words_of(float(value)), dtype=F32))
        return
    call("curla_f64_pack", ptr(value), ptr(words), stream())


def f64_unpack(words, n_mul, n_div, value):
    assert value.dtype == torch.float64 and value.numel() == 1 and words.dtype == F32 and words.numel() == F64_WORDS
    if not words.is_cuda and _lib._trace_hook is None:
        with torch.no_grad():
            value.fill_(f64_of_words(words.tolist(), n_mul, n_div))
        return
    call("curla_f64_unpack", ptr(words), float(n_mul), float(n_div), ptr(value), stream())


def soft_update(param_flat, target_flat, tau):
    # tau and (1 - tau) are rounded to fp32 separately, as `tau * p + (1 - tau) * t` does in torch (utils.py:39-41)
    call("curla_soft_update", ptr(param_flat), ptr(target_flat), param_flat.numel(), float(tau), float(1 - tau),
         stream())


def soft_update2(param_flat, target_flat, split, tau_a, tau_b):
    """One launch for a flat block whose first ``split`` elements use tau_a and the rest tau_b."""
    call("curla_soft_update2", ptr(param_flat), ptr(target_flat), param_flat.numel(), int(split), float(tau_a),
         float(1 - tau_a), float(tau_b), float(1 - tau_b), stream())


def gather_transition_scalars(sc, idx, B, A, act, rew, nd):
    call("curla_gather_transition_scalars", ptr(sc), ptr(idx), B, A, ptr(act), ptr(rew), ptr(nd), stream())


def host_device_pointer(pinned):
    """Device-visible address of a pinned host tensor (raises if the tensor is not pinned / mapped)."""
    import ctypes
    out = ctypes.c_void_p()
    call("curla_host_device_pointer", pinned.data_ptr(), ctypes.addressof(out))
    return out.value


def sample_stage(host_dev_ptr, dev_block, nbytes, sc, B, A, act, rew, nd):
    """gather_transition_scalars with the index block read from pinned host memory (``host_dev_ptr`` from
    host_device_pointer) and written to ``dev_block`` by the same launch."""
    call("curla_sample_stage", host_dev_ptr, ptr(dev_block), nbytes, ptr(sc), B, A, ptr(act), ptr(rew), ptr(nd), stream())


def crop_nchw(frames, idx, h1, w1, B, crop_hw, out_f32=None, out_u8=None):
    _, Hs, Ws, C = frames.shape
    call("curla_crop_nchw", ptr(frames), ptr(idx), ptr(h1), ptr(w1), B, C, Hs, Ws, crop_hw[0], crop_hw[1],
         ptr(out_f32), ptr(out_u8), stream())


def store_frame(chw_u8, frames, slot):
    _, H, W, C = frames.shape
    call("curla_store_frame", ptr(chw_u8), ptr(frames), int(slot), C, H, W, stream())


def gather_stacks(store, fid, idx, B, out):
    """store u8 [F, H, W, 3]; fid int32 [rows, k] (a view with row stride fid.stride(0)); out u8 [B, H, W, 3k]."""
    _, H, W, _ = store.shape
    call("curla_gather_stacks", ptr(store), ptr(fid), fid.stride(0), ptr(idx), B, fid.shape[1], H, W, ptr(out), stream())


def nhwc_to_nchw(x, out):
    B, H, W, C = x.shape
    call("curla_nhwc_to_nchw", ptr(x), ptr(out), B, H, W, C, stream())


def color_jiggle(frames, idx, params, order, B, out):
    _, H, W, C = frames.shape
    call("curla_color_jiggle", ptr(frames), ptr(idx), ptr(params), ptr(order), B, C, H, W, ptr(out), stream())


def noisy_cover(frames, idx, noise, colors, top, bottom, B, out):
    _, H, W, C = frames.shape
    call("curla_noisy_cover", ptr(frames), ptr(idx), ptr(noise), float(colors[0]), float(colors[1]), float(colors[2]),
         int(top), int(bottom), B, C, H, W, ptr(out), stream())


def color_jiggle_nchw(x, params, order, out):
    B, C, H, W = x.shape
    call("curla_color_jiggle_nchw", ptr(_dev(x)), ptr(params), ptr(order), B, C, H, W, ptr(_dev(out)), stream())


def noisy_cover_nchw(x, noise, colors, top, bottom, out):
    B, C, H, W = x.shape
    call("curla_noisy_cover_nchw", ptr(_dev(x)), ptr(_dev(noise)), float(colors[0]), float(colors[1]), float(colors[2]),
         int(top), int(bottom), B, C, H, W, ptr(_dev(out)), stream())


def gather_nhwc(frames, idx, B, out):
    _, H, W, C = frames.shape
    call("curla_gather_nhwc", ptr(frames), ptr(idx), B, C, H, W, ptr(out), stream())
