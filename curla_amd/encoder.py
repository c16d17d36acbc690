"""CNNEncoder / PixelEncoder: the reference's pixel encoder (encoder.py:32-130)
with the arithmetic on the MI355X kernels.

The module tree mirrors the reference (``convs`` ModuleList of Conv2d, ``fc``
Linear, ``ln`` LayerNorm) so ``state_dict()`` keys, ``parameters()`` order and
the RNG stream consumed by construction are the same; the nn modules only hold
Parameters -- ``forward`` runs HIP kernels, never ``nn.Conv2d.forward``.

Differences from the reference, all deliberate:
* ``out_dim`` is computed arithmetically; any H, W >= 7 works (the reference
  table only admits 4 shapes and crashes on its square ones, encoder.py:38-47,66).
* once an encoder is placed on the GPU by ``CurlSacAgent``, ``fc.weight`` stores
  its input columns in (y, x, c) order (NHWC flatten) instead of (c, y, x);
  ``state_dict()`` / ``load_state_dict()`` convert, so checkpoints are
  reference-compatible in both directions.
"""
import os

import torch
import torch.nn as nn

from . import ops


def tie_weights(src, trg):
    """Make ``trg`` use ``src``'s Parameter objects (not copies) for weight and bias (encoder.py:12-15)."""
    if type(src) is not type(trg):
        raise AssertionError("tie_weights needs two layers of the same type")
    trg.weight, trg.bias = src.weight, src.bias


def conv_out_hw(h, w, num_layers):
    h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    for _ in range(num_layers - 1):
        h, w = h - 2, w - 2
    return h, w


class _FcLinear(nn.Linear):
    """nn.Linear whose weight may be held with NHWC-ordered input columns."""

    nhwc = None  # (C, H, W) once the columns are stored in (y, x, c) order

    def to_reference_layout(self, w):
        if self.nhwc is None:
            return w
        c, h, ww = self.nhwc
        return w.reshape(-1, h, ww, c).permute(0, 3, 1, 2).reshape(w.shape[0], -1)

    def from_reference_layout(self, w):
        if self.nhwc is None:
            return w
        c, h, ww = self.nhwc
        return w.reshape(-1, c, h, ww).permute(0, 2, 3, 1).reshape(w.shape[0], -1)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.nhwc is not None:
            destination[prefix + "weight"] = self.to_reference_layout(destination[prefix + "weight"].detach()).contiguous()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "weight"
        if self.nhwc is not None and key in state_dict:
            state_dict = dict(state_dict)
            state_dict[key] = self.from_reference_layout(state_dict[key]).contiguous()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class EncoderWorkspace:
    """Activation buffers of one conv-stack pass for batch size B (NHWC fp32)."""

    def __init__(self, enc, B, device):
        self.B = B
        self.acts = [torch.empty((B, h, w, enc.num_filters), device=device, dtype=torch.float32)
                     for (h, w) in enc.layer_hw[1:]]


class CNNEncoder(nn.Module):
    """Convolutional encoder of pixel observations (encoder.py:32-110)."""

    def __init__(self, obs_shape, feature_dim, num_layers=4, num_filters=32, output_logits=False):
        super().__init__()
        assert len(obs_shape) == 3
        c, h, w = obs_shape
        oh, ow = conv_out_hw(h, w, num_layers)
        if h < 3 or w < 3 or oh < 1 or ow < 1:
            raise NotImplementedError("Encoder does not support input shape")  # encoder.py:46-47
        self.obs_shape = tuple(obs_shape)
        self.feature_dim = feature_dim
        self.num_layers = num_layers
        self.num_filters = num_filters
        self.convs = nn.ModuleList([nn.Conv2d(c, num_filters, 3, stride=2)])
        for _ in range(num_layers - 1):
            self.convs.append(nn.Conv2d(num_filters, num_filters, 3, stride=1))
        # spatial size per layer: index 0 = input, l = output of conv l
        hw = [(h, w), ((h - 3) // 2 + 1, (w - 3) // 2 + 1)]
        for _ in range(num_layers - 1):
            hw.append((hw[-1][0] - 2, hw[-1][1] - 2))
        self.layer_hw = hw
        self.out_dim = hw[-1]
        self.flat_dim = num_filters * hw[-1][0] * hw[-1][1]
        self.fc = _FcLinear(self.flat_dim, feature_dim)
        self.ln = nn.LayerNorm(feature_dim)
        self.outputs = dict()
        self.output_logits = output_logits
        self.record_outputs = False  # fill self.outputs (NCHW copies) for histogram/image logging
        self._ws = {}
        self._partial = {}
        self._nsplit = {}  # batch size -> splits the latest fc_partial of that size wrote

    # -- kernel-side helpers ------------------------------------------------
    def workspace(self, B, tag="infer"):
        key = (B, tag)
        if key not in self._ws:
            self._ws[key] = EncoderWorkspace(self, B, self.fc.weight.device)
        return self._ws[key]

    def conv_params(self):
        return [(m.weight, m.bias) for m in self.convs]

    def ksplit(self, B):
        """Split-K factor of the fc GEMM ([B, flat] x [flat, F]): enough
        workgroups to fill 256 CUs about twice."""
        mt = (B + 63) // 64
        ks = max(1, min(32, 512 // mt))
        return min(ks, max(1, self.flat_dim // 256))

    def partial(self, B):
        """Split-K partial-sum buffer [splits][B][F]: room for the tiled GEMM's ksplit(B) and for the streaming
        kernel's choice (ops.fc_fwd_nsplit, which depends on how many encoders share a launch)."""
        if B not in self._partial:
            ns = max(self.ksplit(B), ops.FC_FWD_MAX_SPLIT if self._streams(B) else 1)
            self._partial[B] = torch.empty((ns, B, self.feature_dim), device=self.fc.weight.device, dtype=torch.float32)
        return self._partial[B]

    def _streams(self, B):
        """Does the fc forward of this batch size take the streaming kernel (curla_fc_fwd_multi)?"""
        return os.environ.get("CURLA_FC_FWD") != "gemm" and ops.fc_fwd_supported(B, self.feature_dim, self.flat_dim)

    def nsplit(self, B):
        """How many splits the partial buffer holds right now (what the latest fc_partial of this batch size wrote)."""
        return self._nsplit.get(B, self.ksplit(B))

    def conv_forward(self, obs_ref, acts, conv_params=None):
        """relu(conv_l(...)) for l = 1..L (encoder.py:77-88); acts[l-1] receives layer l."""
        cp = conv_params or self.conv_params()
        L = self.num_layers
        ops.conv1_fwd(obs_ref, cp[0][0], cp[0][1], acts[0])
        if L > 2 and ops.conv_s1_fwd_stack(acts[0], [cp[i][0] for i in range(1, L)], [cp[i][1] for i in range(1, L)],
                                           acts[1:L]):
            return acts[-1]  # (all stride-1 layers in one launch: batches that are whole rounds of the persistent grid)
        for i in range(1, L):
            ops.conv_s1_fwd(acts[i - 1], cp[i][0], cp[i][1], acts[i])
        return acts[-1]

    def conv_forward2(self, obs_ref, acts, other, other_ref, other_acts):
        """This encoder's conv stack on ``obs_ref`` and ``other``'s (another CNNEncoder of the same geometry, its own
        weights) on ``other_ref``, one launch per layer for both when the kernels can (uint8 handles into one ring);
        otherwise one after the other."""
        cp, cp2 = self.conv_params(), other.conv_params()
        if self.num_layers != other.num_layers or acts[0].shape[1:] != other_acts[0].shape[1:]:
            self.conv_forward(obs_ref, acts)
            other.conv_forward(other_ref, other_acts)
            return
        if ops.conv1_pairable(obs_ref, other_ref):
            ops.conv1_fwd2(obs_ref, cp[0][0], cp[0][1], acts[0], other_ref, cp2[0][0], cp2[0][1], other_acts[0])
        else:  # (float sources, or handles into different rings: the first layer runs twice)
            ops.conv1_fwd(obs_ref, cp[0][0], cp[0][1], acts[0])
            ops.conv1_fwd(other_ref, cp2[0][0], cp2[0][1], other_acts[0])
        L = self.num_layers
        if L > 1 and ops.conv_s1_fwd_stack(acts[0], [cp[i][0] for i in range(1, L)], [cp[i][1] for i in range(1, L)],
                                           acts[1:L], other_acts[0], [cp2[i][0] for i in range(1, L)],
                                           [cp2[i][1] for i in range(1, L)], other_acts[1:L]):
            return  # (all stride-1 layers of both minibatches in one launch)
        for i in range(1, L):
            ops.conv_s1_fwd2(acts[i - 1], cp[i][0], cp[i][1], acts[i], other_acts[i - 1], cp2[i][0], cp2[i][1],
                             other_acts[i])

    def head_forward(self, h, z, fc_out=None, xhat=None, rstd=None, xa=None, act=None):
        """fc + LayerNorm (+tanh) on the NHWC-flattened conv output (encoder.py:98-107).  ``xa`` / ``act``: the
        LayerNorm kernel also writes the Q functions' input rows [z | act] (torch.cat, curl_sac.py:138)."""
        self.fc_partial(h)
        return self.ln_from_partial(h.shape[0], z, fc_out=fc_out, xhat=xhat, rstd=rstd, xa=xa, act=act)

    def fc_partial(self, h):
        """The fc product as split-K partial sums into this encoder's partial buffer (first half of head_forward)."""
        B = h.shape[0]
        if self.fc.nhwc is None:
            raise RuntimeError("encoder weights are not in kernel layout; call CNNEncoder.to_kernel_layout() "
                               "(CurlSacAgent does) before running the HIP path")
        F, K = self.feature_dim, self.flat_dim
        if self._streams(B):
            ns = self._nsplit[B] = ops.fc_fwd_nsplit(1, B, K)
            ops.fc_fwd_multi([h], [self.fc.weight], [self.partial(B)], B, F, K, ns, B * F)
            return
        self._nsplit[B] = self.ksplit(B)
        ops.gemm(h, 0, K, 0, self.fc.weight, 0, K, 0, self.partial(B), F, 0, B, F, K, 1, ksplit=self.ksplit(B),
                 split_stride=B * F)

    @staticmethod
    def fc_partial_multi(pairs):
        """``fc_partial`` of several (encoder, h) pairs of one shape in ONE launch (<= 4): e.g. the three fc layers of
        the critic phase, whose inputs are all ready once the conv launches are through."""
        enc0, h0 = pairs[0]
        B, F, K = h0.shape[0], enc0.feature_dim, enc0.flat_dim
        same = all(e.feature_dim == F and e.flat_dim == K and h.shape[0] == B and e.fc.nhwc is not None for e, h in pairs)
        if not same or len(pairs) > 4 or len(pairs) < 2:
            for e, h in pairs:
                e.fc_partial(h)
            return
        if enc0._streams(B):
            ns = ops.fc_fwd_nsplit(len(pairs), B, K)
            for e, _ in pairs:
                e._nsplit[B] = ns
            ops.fc_fwd_multi([h for _, h in pairs], [e.fc.weight for e, _ in pairs], [e.partial(B) for e, _ in pairs], B, F,
                             K, ns, B * F)
            return
        for e, _ in pairs:
            e._nsplit[B] = enc0.ksplit(B)
        ops.gemm_multi([h for _, h in pairs], [e.fc.weight for e, _ in pairs], [e.partial(B) for e, _ in pairs], B, F, K,
                       ksplit=enc0.ksplit(B), split_stride=B * F)

    def ln_from_partial(self, B, z, fc_out=None, xhat=None, rstd=None, xa=None, act=None):
        """Split-K reduction + bias + LayerNorm (+tanh) of the partial sums left by ``fc_partial``."""
        F = self.feature_dim
        ops.fc_ln_fwd(self.partial(B), self.nsplit(B), B * F, F, self.fc.bias, self.ln.weight, self.ln.bias, B, F, z,
                      fc_out=fc_out, xhat=xhat, rstd=rstd, eps=self.ln.eps, tanh_out=0 if self.output_logits else 1,
                      xa=xa, act=act)
        return z

    @staticmethod
    def ln_from_partial_multi(B, jobs, A=0):
        """``ln_from_partial`` of several encoders of one shape in ONE launch.  jobs = [(encoder, z, kwargs)], kwargs
        as ln_from_partial's (an ``xa`` without ``act`` gets only its feature columns written; A = action columns)."""
        enc0 = jobs[0][0]
        F = enc0.feature_dim
        assert all(e.feature_dim == F and e.nsplit(B) == enc0.nsplit(B) and e.ln.eps == enc0.ln.eps for e, _, _ in jobs)
        ops.fc_ln_fwd_multi([dict(partial=e.partial(B), bias=e.fc.bias, gamma=e.ln.weight, beta=e.ln.bias, y=z,
                                  tanh_out=0 if e.output_logits else 1, **kw) for e, z, kw in jobs],
                            enc0.nsplit(B), B * F, F, B, F, enc0.ln.eps, A)

    def to_kernel_layout(self):
        """Re-order fc.weight's input columns (c,y,x) -> (y,x,c) in place."""
        if self.fc.nhwc is None:
            nhwc = (self.num_filters,) + tuple(self.out_dim)
            with torch.no_grad():
                self.fc.nhwc = nhwc
                self.fc.weight.data.copy_(self.fc.from_reference_layout(self.fc.weight.data.clone()))
        return self

    # -- reference API ---------------------------------------------------------
    def _forward_conv_nhwc(self, obs):
        """The conv stack on the kernels; returns (ObsRef, last activation [B, H, W, C])."""
        ref = obs if isinstance(obs, ops.ObsRef) else ops.ObsRef.from_tensor(obs.contiguous().float())
        ws = self.workspace(ref.B)
        self.conv_forward(ref, ws.acts)
        if self.record_outputs:
            for i, a in enumerate(ws.acts):
                out = torch.empty((a.shape[0], a.shape[3], a.shape[1], a.shape[2]), device=a.device, dtype=a.dtype)
                ops.nhwc_to_nchw(a, out)
                self.outputs["conv%s" % (i + 1)] = out
        return ref, ws.acts[-1]

    def forward_conv(self, obs):
        """encoder.py:77-90: the flattened conv features, in the reference's order -- ``conv.view(B, -1)`` of an NCHW
        tensor, i.e. (c, y, x).  (The kernels keep activations NHWC; this public method transposes a copy.  The
        learner never calls it: ``forward`` feeds ``fc``, whose columns are stored in (y, x, c) order, directly.)"""
        ref, h = self._forward_conv_nhwc(obs)
        out = torch.empty((h.shape[0], h.shape[3], h.shape[1], h.shape[2]), device=h.device, dtype=h.dtype)
        ops.nhwc_to_nchw(h, out)
        return out.view(ref.B, -1)

    def forward(self, obs, detach=False):
        """Inference forward (no autograd graph: training gradients are produced
        by CurlSacAgent's explicit backward kernels).  obs: float NCHW in [0,255]."""
        ref, h = self._forward_conv_nhwc(obs)
        h = h.view(ref.B, -1)
        z = torch.empty((h.shape[0], self.feature_dim), device=h.device, dtype=torch.float32)
        fc_out = torch.empty_like(z) if self.record_outputs else None
        self.head_forward(h, z, fc_out=fc_out)
        if self.record_outputs:
            self.outputs["fc"] = fc_out
            self.outputs["ln" if self.output_logits else "tanh"] = z
        return z

    def copy_conv_weights_from(self, source):
        """Share every conv layer's Parameters with ``source`` (encoder.py:112-116); fc and ln stay separate."""
        for mine, theirs in zip(self.convs, source.convs):
            tie_weights(src=theirs, trg=mine)

    def record_from(self, obs_ref, acts, fc_out, z):
        """Fill ``outputs`` with the reference's entries (encoder.py:79-108) from a pass that has just run through
        the kernels: 'obs' (the /255 input, NCHW), 'conv1'..'convL' (NCHW copies of the NHWC activations), 'fc',
        and 'ln' (or 'tanh').  Called on histogram-logging steps only -- it copies every activation."""
        if obs_ref.is_u8 == 1:
            x = torch.empty((obs_ref.B, obs_ref.C, obs_ref.Hc, obs_ref.Wc), device=z.device, dtype=torch.float32)
            ops.crop_nchw(obs_ref.src, obs_ref.idx, obs_ref.h1, obs_ref.w1, obs_ref.B, (obs_ref.Hc, obs_ref.Wc), out_f32=x)
        elif obs_ref.is_u8 == 2:
            x = torch.empty((obs_ref.B, obs_ref.C, obs_ref.Hc, obs_ref.Wc), device=z.device, dtype=torch.float32)
            ops.nhwc_to_nchw(obs_ref.src, x)
        else:
            x = obs_ref.src.clone()
        self.outputs['obs'] = x.div_(255.0)
        for i, a in enumerate(acts):
            out = torch.empty((a.shape[0], a.shape[3], a.shape[1], a.shape[2]), device=a.device, dtype=a.dtype)
            ops.nhwc_to_nchw(a, out)
            self.outputs['conv%s' % (i + 1)] = out
        self.outputs['fc'] = fc_out.clone()
        self.outputs['ln' if self.output_logits else 'tanh'] = z.clone()

    def log(self, L, step, log_freq):
        """Histograms (and, for feature maps, the first sample as an image) of the recorded outputs plus the
        layer parameters, every ``log_freq`` steps (encoder.py:118-130; same keys)."""
        if step % log_freq:
            return
        for name, value in self.outputs.items():
            L.log_histogram('train_encoder/%s_hist' % name, value, step)
            if value.dim() > 2:
                L.log_image('train_encoder/%s_img' % name, value[0], step)
        for n, conv in enumerate(self.convs, start=1):
            L.log_param('train_encoder/conv%s' % n, conv, step)
        L.log_param('train_encoder/fc', self.fc, step)
        L.log_param('train_encoder/ln', self.ln, step)


PixelEncoder = CNNEncoder  # upstream CURL's name for the same class
