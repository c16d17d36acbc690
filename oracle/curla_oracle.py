"""CPU oracle for the CURL/SAC learner hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch CPU restatement (NumPy for the integer/byte work,
PyTorch-CPU fp32 for the floating-point work) of the path that
``CurlSacAgent.update()`` walks in the reference (paulvantieghem/curla).  It
exists to *check* the HIP path; nothing under ``curla_amd/`` imports it and the
product never falls back to it.  Allowed importers: ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
unmodified reference in the build container (``tests/golden/make_goldens.py``);
``tests/test_oracle_golden.py`` checks every function below against them
(crop/sample indices and cropped bytes bit-exact, losses / activations /
gradients to 1e-5 per-tensor norm-relative).

Every function cites the reference lines it restates (paths relative to the
reference checkout).  The restatement is functional: parameters are a flat
``dict[str, Tensor]`` keyed like the reference ``state_dict`` and all randomness
(crop offsets, sample indices, Gaussian noise) is an explicit input.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# shapes
# --------------------------------------------------------------------------
def conv_out_hw(h: int, w: int, num_layers: int) -> Tuple[int, int]:
    """Spatial size after the conv stack: first layer 3x3 stride 2, the rest
    3x3 stride 1, no padding (encoder.py:54-63).  The reference looks the
    result up in a table (encoder.py:21-29,38-47); this is the arithmetic the
    table rows encode ((76,135)->(31,61), (90,160)->(38,73))."""
    h = (h - 3) // 2 + 1
    w = (w - 3) // 2 + 1
    for _ in range(num_layers - 1):
        h, w = h - 2, w - 2
    return h, w


def random_crop_output_shape(input_hw: Tuple[int, int], factor: float = 0.84) -> Tuple[int, int]:
    """augmentations.py:21-24: ceil(x * 0.84) per side."""
    return tuple(int(np.ceil(x * factor)) for x in input_hw)


# --------------------------------------------------------------------------
# integer / byte path (NumPy)
# --------------------------------------------------------------------------
def draw_sample_cpc_indices(n_valid: int, batch: int, crop_max_h: int, crop_max_w: int,
                            random_crop: bool = True, rng=np.random):
    """The draws ``ReplayBuffer.sample_cpc`` makes from NumPy's global legacy
    stream, in the reference's order (utils.py:147, then augmentations.py:66-67
    once each for obs, next_obs, pos via utils.py:156-158): idxs, (h1,w1) x3.
    ``randint`` upper bounds are exclusive.  Returns (idxs, [(h1,w1)]*3)."""
    idxs = rng.randint(0, n_valid, size=batch)
    offs = []
    if random_crop:
        for _ in range(3):
            h1 = rng.randint(0, crop_max_h, batch)
            w1 = rng.randint(0, crop_max_w, batch)
            offs.append((h1, w1))
    return idxs, offs


def random_crop(imgs: np.ndarray, h1: np.ndarray, w1: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """out[b,c,i,j] = in[b,c,h1[b]+i,w1[b]+j]  (augmentations.py:47-75; the
    reference builds all sliding windows and picks one per sample)."""
    n, c = imgs.shape[:2]
    oh, ow = out_hw
    out = np.empty((n, c, oh, ow), dtype=imgs.dtype)
    for b in range(n):
        out[b] = imgs[b, :, h1[b]:h1[b] + oh, w1[b]:w1[b] + ow]
    return out


def center_crop(img: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """augmentations.py:26-45 (evaluation augmentation of RandomCrop)."""
    h, w = img.shape[-2:]
    oh, ow = out_hw
    top, left = (h - oh) // 2, (w - ow) // 2
    return img[..., top:top + oh, left:left + ow]


# --------------------------------------------------------------------------
# float augmentations
# --------------------------------------------------------------------------
def noisy_cover(imgs: torch.Tensor, colors, noise: torch.Tensor, top_ratio=0.31, bottom_ratio=0.20) -> torch.Tensor:
    """NoisyCover.training_augmentation (augmentations.py:138-205) with the noise
    tensor explicit: rows [0, ceil(.31 h)) and [h - ceil(.20 h), h) of every RGB
    frame get colors[c % 3]; + noise; clamp to [0, 255].  imgs: float [B, C, H, W].
    The cover logic is the reference's own code (pinned by noisy_cover.npz); the
    noise distribution is kornia's RandomGaussianNoise(std=10) there."""
    h = imgs.shape[2]
    top, bottom = int(np.ceil(h * top_ratio)), int(np.ceil(h * bottom_ratio))
    out = imgs.clone().float()
    rows = list(range(0, top)) + list(range(h - bottom, h))
    for c in range(out.shape[1]):
        out[:, c, rows, :] = float(colors[c % 3])
    return torch.clamp(out + noise, 0, 255)


def _rgb_to_hsv(img: torch.Tensor):
    """kornia-style RGB->HSV on [..., 3, H, W] in [0,1]: h in [0, 2pi)."""
    r, g, b = img[..., 0, :, :], img[..., 1, :, :], img[..., 2, :, :]
    mx, arg = img.max(-3)
    mn = img.min(-3)[0]
    v = mx
    d = mx - mn
    s = d / (mx + 1e-8)
    d = torch.where(d == 0, torch.ones_like(d), d)
    rc, gc, bc = mx - r, mx - g, mx - b
    h = torch.stack([bc - gc, (rc - bc) + 2.0 * d, (gc - rc) + 4.0 * d], -3)
    h = torch.gather(h, -3, arg.unsqueeze(-3)).squeeze(-3) / d
    h = (h / 6.0) % 1.0
    return 2.0 * math.pi * h, s, v


def _hsv_to_rgb(h, s, v):
    h6 = h / (2.0 * math.pi) * 6.0
    hi = torch.floor(h6) % 6
    f = (h6 % 6) - hi
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    hi = hi.long()
    sel = lambda a0, a1, a2, a3, a4, a5: torch.stack([a0, a1, a2, a3, a4, a5], 0).gather(0, hi.unsqueeze(0)).squeeze(0)  # noqa: E731
    return torch.stack([sel(v, q, p, p, t, v), sel(t, v, v, q, p, p), sel(p, p, t, v, v, q)], -3)


def color_jiggle(imgs_u8: np.ndarray, params: torch.Tensor, order) -> torch.Tensor:
    """ColorJiggle.training_augmentation (augmentations.py:78-136) -- PARITY UNPINNED:
    the reference calls kornia.augmentation.ColorJiggle(brightness 0, contrast .2,
    saturation .5, hue .5, p .85), whose source is not part of the reference and whose
    version is not pinned.  This restates kornia's documented behaviour: each RGB
    frame of the stack is one image; with probability p the four operations run in
    the (per call) random ``order``: 0 brightness (additive, factor 0 -> identity),
    1 contrast (x*c, clamp [0,1]), 2 saturation (HSV, s*f clamp [0,1]), 3 hue (HSV,
    h + d mod 2pi).  params [B*k, 4] = (apply, contrast, saturation, hue_rad).
    imgs_u8 [B, C, H, W] -> float [B, C, H, W] in [0, 255]."""
    x = torch.from_numpy(np.ascontiguousarray(imgs_u8)).float() / 255.0
    B, C, H, W = x.shape
    x = x.reshape(B * (C // 3), 3, H, W)
    out = x.clone()
    for i in range(x.shape[0]):
        if params[i, 0] == 0:
            continue
        im = x[i]
        for op in [int(o) for o in order]:
            if op == 1:
                im = torch.clamp(im * params[i, 1], 0, 1)
            elif op == 2:
                h, s_, v = _rgb_to_hsv(im)
                im = _hsv_to_rgb(h, torch.clamp(s_ * params[i, 2], 0, 1), v)
            elif op == 3:
                h, s_, v = _rgb_to_hsv(im)
                im = _hsv_to_rgb(torch.fmod(torch.fmod(h + params[i, 3], 2 * math.pi) + 2 * math.pi, 2 * math.pi), s_, v)
        out[i] = im
    return (out * 255.0).reshape(B, C, H, W)


# --------------------------------------------------------------------------
# networks (PyTorch CPU fp32)
# --------------------------------------------------------------------------
class _ReluGivenBranch(torch.autograd.Function):
    """relu(x) whose DERIVATIVE takes its branch (x > 0 or not) from a given boolean tensor instead of from x.
    The value is the reference's ``torch.relu`` (encoder.py:81,86) unchanged.  Used by the full-size parity tests
    only: a pre-activation within rounding of 0 is positive in one fp32 evaluation of the network and not in
    another (1-2 elements out of 20 million per layer at B=512), and since a conv weight gradient is a sum of
    ~600k signed, largely cancelling terms, ONE such element moves it by ~1e-4 .. 1e-3 of its size.  Comparing two
    evaluations element by element is meaningful only when both took the same branch at those elements; the tests
    assert separately that the disagreeing elements are few and all within rounding of 0.

    The branch is looked up at BACKWARD time in ``plan[key]`` (None / missing: the ReLU's own x > 0), so one forward
    pass can be differentiated several times under different branch decisions (``regrad`` of the phase functions)."""

    @staticmethod
    def forward(ctx, x, plan, key):
        ctx.plan, ctx.key = plan, key
        ctx.save_for_backward(x > 0)
        return x.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        positive = ctx.plan.get(ctx.key)
        if positive is None:
            (positive,) = ctx.saved_tensors
        return g * positive.to(g.dtype), None, None


def _relu(x, plan, key):
    """torch.relu, or -- when the caller differentiates along given branches (``plan`` is a dict) -- the same values
    with the derivative's branch taken from ``plan[key]``."""
    return torch.relu(x) if plan is None else _ReluGivenBranch.apply(x, plan, key)


def _branch_plan(relu_branches=None, q_branches=None, trunk_branches=None, force=False):
    """The dict _ReluGivenBranch reads: ("conv", i) -> [B, C, H, W] booleans, ("Q1." | "Q2." | "trunk.", j) ->
    [B, hidden] booleans.  None when no branch is given (plain torch.relu everywhere) unless ``force``."""
    if relu_branches is None and q_branches is None and trunk_branches is None and not force:
        return None
    plan = {}
    for i, m in enumerate(relu_branches or []):
        plan[("conv", i)] = m
    for name, pair in zip(("Q1.trunk.", "Q2.trunk."), q_branches or []):
        for j, m in enumerate(pair or []):
            plan[(name, j)] = m
    for j, m in enumerate(trunk_branches or []):
        plan[("trunk.", j)] = m
    return plan


def _conv2d(x, w, b, stride):
    """F.conv2d (encoder.py:80,85).  float64 inputs (the arbiter passes of tests/test_gpu_fullsize.py) go through
    PyTorch's im2col path, whose column buffer covers the whole batch (15 GB for one layer of configs[4]): those are
    evaluated 64 samples at a time -- a sample's result does not depend on its batch."""
    if x.dtype != torch.float64 or x.shape[0] <= 64 or x.is_cuda:
        return F.conv2d(x, w, b, stride=stride)
    return torch.cat([F.conv2d(xc, w, b, stride=stride) for xc in x.split(64)])


def as_dtype(x, dtype, device=None):
    """Tensors / dicts / lists of tensors cast to ``dtype`` (detached copies; None and non-float tensors pass
    through): the float64 evaluation of a phase is the same function on ``as_dtype(..., torch.float64)`` arguments.
    ``device`` (tests only): also move them there -- the full-size float64 arbiter lets PyTorch's own float64 kernels on
    the GPU evaluate these same functions (240 s -> 15 s for configs[4]; pinned to the host evaluation by the test)."""
    if x is None:
        return None
    if isinstance(x, dict):
        return {k: as_dtype(v, dtype, device) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(as_dtype(v, dtype, device) for v in x)
    if torch.is_tensor(x) and x.is_floating_point():
        return x.detach().to(device=device, dtype=dtype) if device is not None else x.detach().to(dtype)
    if torch.is_tensor(x) and device is not None:
        return x.to(device)
    return x


def encoder_forward(p: Params, prefix: str, obs: torch.Tensor, num_layers: int,
                    detach: bool = False, output_logits: bool = True,
                    conv_prefix: Optional[str] = None, outputs: Optional[dict] = None,
                    plan: Optional[dict] = None) -> torch.Tensor:
    """CNNEncoder.forward (encoder.py:77-110).  obs: float NCHW in [0,255].
    ``conv_prefix`` lets the actor use the critic's conv tensors (weight tying,
    encoder.py:112-116 / curl_sac.py:290).  ``plan`` (tests only, see _branch_plan): the branch each conv ReLU's
    derivative takes; None = the plain ``torch.relu``."""
    cp = conv_prefix if conv_prefix is not None else prefix
    x = obs / 255.0
    for i in range(num_layers):
        x = _conv2d(x, p[f"{cp}convs.{i}.weight"], p[f"{cp}convs.{i}.bias"], 2 if i == 0 else 1)
        x = _relu(x, plan, ("conv", i))
        if outputs is not None:
            outputs[f"conv{i + 1}"] = x
    h = x.reshape(x.size(0), -1)
    if detach:
        h = h.detach()
    h_fc = F.linear(h, p[f"{prefix}fc.weight"], p[f"{prefix}fc.bias"])
    h_norm = F.layer_norm(h_fc, (h_fc.size(-1),), p[f"{prefix}ln.weight"], p[f"{prefix}ln.bias"], 1e-5)
    if outputs is not None:
        outputs["fc"] = h_fc
        outputs["ln"] = h_norm
    return h_norm if output_logits else torch.tanh(h_norm)


def mlp3(p: Params, prefix: str, x: torch.Tensor, plan: Optional[dict] = None,
         hidden: Optional[list] = None) -> torch.Tensor:
    """Linear-ReLU-Linear-ReLU-Linear trunk (curl_sac.py:70-74,129-133).  ``plan`` (tests only, see _branch_plan):
    the branch each of the two ReLUs' derivatives takes (a hidden unit within rounding of 0 moves a trunk weight
    gradient by ~1/sqrt(B) of a row's size).  ``hidden`` (tests only): a list that receives the two PRE-activations
    [B, hidden] (detached)."""
    h1 = F.linear(x, p[f"{prefix}0.weight"], p[f"{prefix}0.bias"])
    h2 = F.linear(_relu(h1, plan, (prefix, 0)), p[f"{prefix}2.weight"], p[f"{prefix}2.bias"])
    if hidden is not None:
        hidden += [h1.detach(), h2.detach()]
    return F.linear(_relu(h2, plan, (prefix, 1)), p[f"{prefix}4.weight"], p[f"{prefix}4.bias"])


def actor_forward(actor: Params, critic: Params, obs: torch.Tensor, noise: Optional[torch.Tensor],
                  num_layers: int, log_std_min: float, log_std_max: float,
                  detach_encoder: bool = False, compute_pi: bool = True, compute_log_pi: bool = True,
                  plan: Optional[dict] = None, hidden: Optional[list] = None):
    """Actor.forward + gaussian_logprob + squash (curl_sac.py:20-35,79-110).
    The conv tensors come from ``critic`` (tied); fc/ln/trunk from ``actor``.
    ``noise`` replaces ``torch.randn_like(mu)`` (curl_sac.py:97)."""
    merged = dict(actor)
    for k, v in critic.items():
        if k.startswith("encoder.convs."):
            merged[k] = v
    z = encoder_forward(merged, "encoder.", obs, num_layers, detach=detach_encoder)
    mu, log_std = mlp3(merged, "trunk.", z, plan, hidden).chunk(2, dim=-1)
    log_std = torch.tanh(log_std)
    log_std = log_std_min + 0.5 * (log_std_max - log_std_min) * (log_std + 1)
    pi = log_pi = None
    if compute_pi:
        pi = mu + noise * log_std.exp()
    if compute_log_pi:
        residual = (-0.5 * noise.pow(2) - log_std).sum(-1, keepdim=True)
        log_pi = residual - 0.5 * np.log(2 * np.pi) * noise.size(-1)
    mu = torch.tanh(mu)
    if pi is not None:
        pi = torch.tanh(pi)
    if log_pi is not None:
        log_pi = log_pi - torch.log(F.relu(1 - pi.pow(2)) + 1e-6).sum(-1, keepdim=True)
    return mu, pi, log_pi, log_std


def critic_forward(critic: Params, obs: torch.Tensor, action: torch.Tensor, num_layers: int,
                   detach_encoder: bool = False, outputs: Optional[dict] = None, plan: Optional[dict] = None,
                   hidden: Optional[list] = None):
    """Critic.forward / QFunction.forward (curl_sac.py:135-169).  ``plan`` (tests only): see _branch_plan;
    ``hidden`` (tests only) receives Q1's then Q2's two hidden pre-activations."""
    z = encoder_forward(critic, "encoder.", obs, num_layers, detach=detach_encoder, outputs=outputs, plan=plan)
    za = torch.cat([z, action], dim=1)
    return mlp3(critic, "Q1.trunk.", za, plan, hidden), mlp3(critic, "Q2.trunk.", za, plan, hidden)


def curl_logits(W: torch.Tensor, z_a: torch.Tensor, z_pos: torch.Tensor) -> torch.Tensor:
    """CURL.compute_logits (curl_sac.py:211-222)."""
    Wz = torch.matmul(W, z_pos.T)
    logits = torch.matmul(z_a, Wz)
    return logits - torch.max(logits, 1)[0][:, None]


# --------------------------------------------------------------------------
# phases: loss + gradients from explicit weights / batch / noise
# --------------------------------------------------------------------------
def _leafify(p: Params) -> Params:
    return {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}


def _grads(p: Params) -> Dict[str, Optional[torch.Tensor]]:
    return {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in p.items()}


def _regrad_fn(plan: dict, leaves, backward, collect):
    """``regrad(relu_branches=None, q_branches=None, trunk_branches=None)`` of a phase evaluated with ``regrad=True``:
    differentiates the SAME forward pass again with the ReLU derivatives along the given branches (None: every ReLU's
    own) and returns what ``collect()`` gathers.  Values are untouched by construction: only backward passes rerun."""
    def regrad(relu_branches=None, q_branches=None, trunk_branches=None):
        plan.clear()
        plan.update(_branch_plan(relu_branches, q_branches, trunk_branches, force=True))
        for v in leaves:
            v.grad = None
        backward(True)
        return collect()
    return regrad


def critic_phase(actor: Params, critic: Params, critic_target: Params, log_alpha: torch.Tensor,
                 obs, action, reward, next_obs, not_done, noise, *, num_layers: int, discount: float,
                 log_std_min: float, log_std_max: float, detach_encoder: bool = False,
                 relu_branches: Optional[list] = None, q_branches: Optional[list] = None, regrad: bool = False):
    """CurlSacAgent.update_critic up to and including backward
    (curl_sac.py:349-367).  Returns dict(loss, target_Q, q1, q2, grads, enc).
    ``relu_branches`` / ``q_branches`` (tests only): the derivative branches of the conv ReLUs (one boolean NCHW tensor
    per layer) and of the twin-Q hidden units ([Q1's pair, Q2's pair] of [B, hidden] booleans) in the critic's
    differentiated pass over ``obs``; ``regrad=True`` adds ``regrad(...)`` to the result (see _regrad_fn) -> grads."""
    with torch.no_grad():
        _, policy_action, log_pi, _ = actor_forward(actor, critic, next_obs, noise, num_layers,
                                                    log_std_min, log_std_max)
        tq1, tq2 = critic_forward(critic_target, next_obs, policy_action, num_layers)
        alpha = log_alpha.detach().exp()
        target_V = torch.min(tq1, tq2) - alpha * log_pi
        target_Q = reward + (not_done * discount * target_V)
        # curl_sac.py:353-355 has no cast: `self.alpha` is a 0-dim float64 tensor, and a 0-dim operand does not promote
        # a float32 tensor, so the reference's target_Q is float32.  The cast below is therefore a no-op in the fp32
        # evaluation; it only keeps target_Q in the ARGUMENTS' dtype when the tests evaluate this function in float64.
        target_Q = target_Q.to(reward.dtype)
    c = _leafify(critic)
    enc, qh = {}, []
    plan = _branch_plan(relu_branches, q_branches, force=regrad)
    q1, q2 = critic_forward(c, obs, action, num_layers, detach_encoder=detach_encoder, outputs=enc, plan=plan, hidden=qh)
    loss = F.mse_loss(q1, target_Q) + F.mse_loss(q2, target_Q)
    loss.backward(retain_graph=regrad)
    out = dict(loss=loss.detach(), target_Q=target_Q, q1=q1.detach(), q2=q2.detach(),
               policy_action=policy_action, next_log_pi=log_pi,
               grads=_grads(c), enc={k: v.detach() for k, v in enc.items()}, q_hidden=qh)
    if regrad:
        out["regrad"] = _regrad_fn(plan, list(c.values()), lambda keep: loss.backward(retain_graph=keep), lambda: _grads(c))
    return out


def actor_phase(actor: Params, critic: Params, log_alpha: torch.Tensor, obs, noise, *,
                num_layers: int, log_std_min: float, log_std_max: float, target_entropy: float,
                trunk_branches: Optional[list] = None, q_branches: Optional[list] = None, regrad: bool = False):
    """CurlSacAgent.update_actor_and_alpha up to the two backward calls
    (curl_sac.py:373-403).  Live gradients: the actor's own fc/ln/trunk and
    log_alpha; gradients deposited on critic tensors are dead in the reference
    (cleared before any step reads them) and are not returned.  ``trunk_branches`` / ``q_branches`` / ``regrad``
    (tests only): as in critic_phase; ``regrad(...)`` returns the actor's live gradients."""
    a = _leafify(actor)
    la = log_alpha.detach().clone().requires_grad_(True)
    th, qh = [], []
    plan = _branch_plan(None, q_branches, trunk_branches, force=regrad)
    _, pi, log_pi, log_std = actor_forward(a, critic, obs, noise, num_layers, log_std_min, log_std_max,
                                           detach_encoder=True, plan=plan, hidden=th)
    q1, q2 = critic_forward(critic, obs, pi, num_layers, detach_encoder=True, plan=plan, hidden=qh)
    actor_Q = torch.min(q1, q2)
    actor_loss = (la.exp().detach() * log_pi - actor_Q).mean()
    entropy = 0.5 * log_std.shape[1] * (1.0 + np.log(2 * np.pi)) + log_std.sum(dim=-1)
    actor_loss.backward(retain_graph=regrad)
    alpha_loss = (la.exp() * (-log_pi - target_entropy).detach()).mean()
    alpha_loss.backward()
    live = lambda: {k: g for k, g in _grads(a).items() if g is not None}  # noqa: E731
    out = dict(actor_loss=actor_loss.detach(), alpha_loss=alpha_loss.detach(), entropy=entropy.mean().detach(),
               alpha=la.exp().detach(), pi=pi.detach(), log_pi=log_pi.detach(), log_std=log_std.detach(),
               q1=q1.detach(), q2=q2.detach(), grads=live(), log_alpha_grad=la.grad.detach().clone(),
               trunk_hidden=th, q_hidden=qh)
    if regrad:
        out["regrad"] = _regrad_fn(plan, list(a.values()), lambda keep: actor_loss.backward(retain_graph=keep), live)
    return out


def cpc_phase(critic: Params, critic_target: Params, W: torch.Tensor, obs_anchor, obs_pos, *, num_layers: int,
              relu_branches: Optional[list] = None, regrad: bool = False):
    """CurlSacAgent.update_cpc up to backward (curl_sac.py:406-417): anchors
    through the online encoder, positives through the target encoder under
    no_grad, bilinear logits, cross-entropy against arange(B).  ``relu_branches`` / ``regrad`` (tests only): as in
    critic_phase; ``regrad(...)`` returns (encoder grads, W grad)."""
    enc = _leafify({k: v for k, v in critic.items() if k.startswith("encoder.")})
    Wl = W.detach().clone().requires_grad_(True)
    plan = _branch_plan(relu_branches, force=regrad)
    acts = {}
    z_a = encoder_forward(enc, "encoder.", obs_anchor, num_layers, plan=plan, outputs=acts)
    with torch.no_grad():
        z_pos = encoder_forward(critic_target, "encoder.", obs_pos, num_layers)
    logits = curl_logits(Wl, z_a, z_pos)
    labels = torch.arange(logits.shape[0], device=logits.device).long()
    loss = F.cross_entropy(logits, labels)
    loss.backward(retain_graph=regrad)
    out = dict(loss=loss.detach(), z_a=z_a.detach(), z_pos=z_pos, logits=logits.detach(),
               grads=_grads(enc), W_grad=Wl.grad.detach().clone(), enc={k: v.detach() for k, v in acts.items()})
    if regrad:
        out["regrad"] = _regrad_fn(plan, list(enc.values()) + [Wl], lambda keep: loss.backward(retain_graph=keep),
                                   lambda: (_grads(enc), Wl.grad.detach().clone()))
    return out


def soft_update(net: Params, target: Params, tau: float, prefix: str) -> None:
    """utils.soft_update_params (utils.py:37-41), in place on ``target`` for
    every tensor under ``prefix``."""
    for k, v in net.items():
        if k.startswith(prefix):
            target[k].copy_(tau * v + (1 - tau) * target[k])


# --------------------------------------------------------------------------
# initialisation (for the CPU baseline / smoke; parity tests load fixtures)
# --------------------------------------------------------------------------
def _orthogonal(shape, gain=1.0, gen=None):
    t = torch.empty(shape)
    torch.nn.init.orthogonal_(t, gain, generator=gen)
    return t


def init_encoder(prefix: str, in_ch: int, hw: Tuple[int, int], feature_dim: int, num_layers: int,
                 num_filters: int, gen=None) -> Params:
    """Shapes of CNNEncoder.__init__ (encoder.py:54-67) with weight_init
    (curl_sac.py:38-54): delta-orthogonal convs, orthogonal fc, zero biases,
    LayerNorm ones/zeros."""
    p: Params = {}
    gain = math.sqrt(2.0)
    for i in range(num_layers):
        cin = in_ch if i == 0 else num_filters
        w = torch.zeros(num_filters, cin, 3, 3)
        w[:, :, 1, 1] = _orthogonal((num_filters, cin), gain, gen)
        p[f"{prefix}convs.{i}.weight"] = w
        p[f"{prefix}convs.{i}.bias"] = torch.zeros(num_filters)
    oh, ow = conv_out_hw(hw[0], hw[1], num_layers)
    p[f"{prefix}fc.weight"] = _orthogonal((feature_dim, num_filters * oh * ow), 1.0, gen)
    p[f"{prefix}fc.bias"] = torch.zeros(feature_dim)
    p[f"{prefix}ln.weight"] = torch.ones(feature_dim)
    p[f"{prefix}ln.bias"] = torch.zeros(feature_dim)
    return p


def init_mlp3(prefix: str, din: int, hidden: int, dout: int, gen=None) -> Params:
    p: Params = {}
    for j, (a, b) in zip((0, 2, 4), ((din, hidden), (hidden, hidden), (hidden, dout))):
        p[f"{prefix}{j}.weight"] = _orthogonal((b, a), 1.0, gen)
        p[f"{prefix}{j}.bias"] = torch.zeros(b)
    return p


class OracleAgent:
    """Stateful wrapper chaining the phases with ``torch.optim.Adam`` exactly as
    ``CurlSacAgent.__init__/update`` does (curl_sac.py:226-318,426-451): five
    optimizers, tied convs, the encoder stepped by both ``encoder_optimizer``
    and ``cpc_optimizer`` from the same gradients.  Used for the CPU baseline
    timing and for the smoke check; parity is asserted per phase."""

    def __init__(self, obs_shape, action_shape, hidden_dim=1024, discount=0.99, init_temperature=0.1,
                 alpha_lr=1e-4, alpha_beta=0.5, actor_lr=1e-3, actor_beta=0.9, actor_log_std_min=-10,
                 actor_log_std_max=2, actor_update_freq=2, critic_lr=1e-3, critic_beta=0.9, critic_tau=0.01,
                 critic_target_update_freq=2, encoder_feature_dim=50, encoder_lr=1e-3, encoder_tau=0.05,
                 num_layers=4, num_filters=32, cpc_update_freq=1, detach_encoder=False, pixel_sac=False,
                 seed=1):
        gen = torch.Generator().manual_seed(seed)
        c, h, w = obs_shape
        A = action_shape[0]
        self.num_layers, self.discount = num_layers, discount
        self.lo, self.hi = actor_log_std_min, actor_log_std_max
        self.actor_update_freq, self.critic_target_update_freq = actor_update_freq, critic_target_update_freq
        self.cpc_update_freq, self.critic_tau, self.encoder_tau = cpc_update_freq, critic_tau, encoder_tau
        self.detach_encoder, self.pixel_sac = detach_encoder, pixel_sac
        self.target_entropy = -float(np.prod(action_shape))
        self.critic = init_encoder("encoder.", c, (h, w), encoder_feature_dim, num_layers, num_filters, gen)
        self.critic.update(init_mlp3("Q1.trunk.", encoder_feature_dim + A, hidden_dim, 1, gen))
        self.critic.update(init_mlp3("Q2.trunk.", encoder_feature_dim + A, hidden_dim, 1, gen))
        self.critic_target = {k: v.clone() for k, v in self.critic.items()}
        enc_a = init_encoder("encoder.", c, (h, w), encoder_feature_dim, num_layers, num_filters, gen)
        self.actor = {k: v for k, v in enc_a.items() if ".convs." not in k}
        self.actor.update(init_mlp3("trunk.", encoder_feature_dim, hidden_dim, 2 * A, gen))
        self.W = torch.rand(encoder_feature_dim, encoder_feature_dim, generator=gen)
        self.log_alpha = torch.tensor(np.log(init_temperature))
        for d in (self.critic, self.actor):
            for v in d.values():
                v.requires_grad_(True)
        self.W.requires_grad_(True)
        self.log_alpha.requires_grad_(True)
        enc_params = [v for k, v in self.critic.items() if k.startswith("encoder.")]
        self.actor_opt = torch.optim.Adam(list(self.actor.values()), lr=actor_lr, betas=(actor_beta, 0.999))
        self.critic_opt = torch.optim.Adam(list(self.critic.values()), lr=critic_lr, betas=(critic_beta, 0.999))
        self.alpha_opt = torch.optim.Adam([self.log_alpha], lr=alpha_lr, betas=(alpha_beta, 0.999))
        self.encoder_opt = torch.optim.Adam(enc_params, lr=encoder_lr)
        self.cpc_opt = torch.optim.Adam([self.W] + enc_params, lr=encoder_lr)

    @staticmethod
    def _apply(opt, params, grads):
        for v, g in zip(params, grads):
            v.grad = g
        opt.step()
        for v in params:
            v.grad = None

    def update(self, obs, action, reward, next_obs, not_done, pos, noise_c, noise_a, step, only_cpc=False):
        out = {}
        if not only_cpc:
            r = critic_phase(self.actor, self.critic, self.critic_target, self.log_alpha, obs, action, reward,
                             next_obs, not_done, noise_c, num_layers=self.num_layers, discount=self.discount,
                             log_std_min=self.lo, log_std_max=self.hi, detach_encoder=self.detach_encoder)
            out["critic_loss"] = r["loss"]
            keys = list(self.critic.keys())
            self._apply(self.critic_opt, [self.critic[k] for k in keys], [r["grads"][k] for k in keys])
            if step % self.actor_update_freq == 0:
                r = actor_phase(self.actor, self.critic, self.log_alpha, obs, noise_a, num_layers=self.num_layers,
                                log_std_min=self.lo, log_std_max=self.hi, target_entropy=self.target_entropy)
                out["actor_loss"], out["alpha_loss"] = r["actor_loss"], r["alpha_loss"]
                keys = list(self.actor.keys())
                self._apply(self.actor_opt, [self.actor[k] for k in keys], [r["grads"][k] for k in keys])
                self._apply(self.alpha_opt, [self.log_alpha], [r["log_alpha_grad"]])
            if step % self.critic_target_update_freq == 0:
                with torch.no_grad():
                    soft_update(self.critic, self.critic_target, self.critic_tau, "Q1.")
                    soft_update(self.critic, self.critic_target, self.critic_tau, "Q2.")
                    soft_update(self.critic, self.critic_target, self.encoder_tau, "encoder.")
        if not self.pixel_sac and step % self.cpc_update_freq == 0:
            r = cpc_phase(self.critic, self.critic_target, self.W, obs, pos, num_layers=self.num_layers)
            out["curl_loss"] = r["loss"]
            keys = [k for k in self.critic if k.startswith("encoder.")]
            params = [self.critic[k] for k in keys]
            grads = [r["grads"][k] for k in keys]
            self._apply(self.encoder_opt, params, grads)
            self._apply(self.cpc_opt, [self.W] + params, [r["W_grad"]] + grads)
        return out
