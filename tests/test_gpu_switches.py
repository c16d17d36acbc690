"""Every kernel-selection switch of the library under ONE whole-update parity test.

The product library carries alternatives to its default paths: fallbacks for shapes the default does not take and A/B
partners for measurements.  Code that only an environment variable can reach is code nobody tests, so (a) the C-level
switches are run-time options (``curla_set_option``, curla_amd/csrc/options.h -- no ``getenv`` after start-up) and (b)
every value of every switch, C-level and Python-level, runs here: one even-step ``agent.update()`` (critic phase with
the fused target lerp, actor / alpha phase, CURL phase: curl_sac.py:426-451) at the smallest batch that takes every
fused path (B = 256: a multiple of the CU count and of 128), on the same weights, minibatch and noise as the default
path, all learning rates zero so that every phase sees the same weights on both sides.  Compared per tensor at 1e-4
(max|a-b| / max|b|): the five logged losses and every gradient an optimizer consumes (24 critic + 10 actor + log_alpha
+ 12 encoder + W).  The default path itself is held to the oracle at this size by
tests/test_gpu_agent.py::test_one_update_at_batch_256_takes_the_fused_paths_and_matches_the_oracle.
Where a switch is visible at the C ABI (another entry point is called), the call trace must show it."""
import collections
import contextlib
import os

import numpy as np
import pytest
import torch

from tests._util import RTOL, rel_err
from tests.test_gpu_agent import HP, NullLogger, grads_of

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def _switched(options, env):
    from curla_amd import _lib
    old_env = {k: os.environ.get(k) for k in env}
    with contextlib.ExitStack() as stack:
        for k, v in options.items():
            stack.enter_context(_lib.option(k, v))
        os.environ.update(env)
        try:
            yield
        finally:
            for k, v in old_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


def _one_update(aug_name, options=None, env=None, lr=0.0, steps=1):
    """One even-step update() (``steps`` = 2 adds the odd step) of a seeded agent on a seeded ring under the given
    switches.  Returns (losses, gradients captured in front of each optimizer step, C-ABI call counts, parameters)."""
    import curla_amd
    import curla_amd.ops as ops_mod
    import curla_amd.optim as optim_mod
    from curla_amd import _lib
    with _switched(options or {}, env or {}):
        torch.manual_seed(11)
        np.random.seed(11)
        dev = torch.device("cuda")
        B, hidden, layers = 256, 64, 4
        if aug_name == "random_crop":
            C, in_hw, out_hw = 9, (40, 44), (32, 36)
            aug = curla_amd.RandomCrop(in_hw, out_hw)
        elif aug_name == "wide":  # rows of >= 16 pixel quads after three stride-1 layers: the F(4,3) forward by default
            C, in_hw, out_hw, layers = 9, (20, 150), (16, 141), 4
            aug = curla_amd.RandomCrop(in_hw, out_hw)
        else:  # float NHWC minibatches: the colour-jittered observations of configs[4]
            C, in_hw, out_hw = 12, (36, 40), (36, 40)
            aug = curla_amd.ColorJiggle(in_hw)
        hp = {**HP, "num_layers": layers, "alpha_lr": lr * 0.1, "actor_lr": lr, "critic_lr": lr, "encoder_lr": lr}
        agent = curla_amd.CurlSacAgent((C,) + out_hw, (2,), dev, aug, hidden_dim=hidden, **hp)
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():  # all nine taps of every conv live, biases off zero (the init is centre-tap only)
            sd = {k: v.detach().cpu().clone() for k, v in agent.critic.state_dict().items()}
            for k in sd:
                if ".convs." in k:
                    sd[k] += 0.05 * torch.randn(sd[k].shape, generator=g)
            agent.critic.load_state_dict(sd)
            agent.critic_target.load_state_dict({k: 0.98 * v for k, v in sd.items()})
            agent.actor.load_state_dict({**{k: v.detach().cpu().clone() for k, v in agent.actor.state_dict().items()},
                                         **{k: v for k, v in sd.items() if ".convs." in k}})
        rb = curla_amd.ReplayBuffer((C,) + in_hw, (2,), 512, B, dev, aug)
        rs = np.random.RandomState(6)
        n = 300
        rb.add_batch(rs.randint(0, 256, (n, C) + in_hw, dtype=np.uint8), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                     rs.randn(n).astype(np.float32), rs.randint(0, 256, (n, C) + in_hw, dtype=np.uint8),
                     (np.arange(n) % 7) == 6)
        noises = iter([torch.randn(B, 2, generator=g).cuda() for _ in range(4)])
        agent._noise = lambda ws, noise: (ws.noise.copy_(next(noises)), None)
        grads = {}

        def hook(opt, name, collect):
            real = opt.step

            def step(*a, **k):
                grads.setdefault(name, collect())
                return real(*a, **k)
            opt.step = step
        hook(agent.critic_optimizer, "critic", lambda: grads_of(agent.critic))
        hook(agent.actor_optimizer, "actor",
             lambda: {**{k: v for k, v in grads_of(agent.actor).items() if ".convs." not in k},
                      "log_alpha": agent.log_alpha.grad.detach().cpu().reshape(1).clone()})
        if hasattr(agent, "encoder_optimizer"):
            hook(agent.encoder_optimizer, "cpc", lambda: {**grads_of(agent.critic.encoder, "encoder."),
                                                           "W": agent.CURL.W.grad.detach().cpu().clone()})
        calls = collections.Counter()
        real_call = _lib.call

        def traced(name, *a):
            calls[name] += 1
            return real_call(name, *a)
        mods = (ops_mod, optim_mod)
        for m in mods:
            m.call = traced
        L = NullLogger()
        try:
            for step in range(steps):
                agent.update(rb, L, step)
            torch.cuda.synchronize()
        finally:
            for m in mods:
                m.call = real_call
        params = {"critic": agent._critic_flat.detach().cpu().clone(), "actor": agent._actor_flat.detach().cpu().clone(),
                  "target": agent._target_flat.detach().cpu().clone()}
        # the online conv activations of obs (their signs are the ReLU branches the conv gradients went along)
        params["acts"] = [a.detach().clone() for a in agent._ws(B).acts_main[:layers]]
        return dict(L.scalars), grads, calls, params


_DEFAULT = {}


def _default(aug_name):
    if aug_name not in _DEFAULT:
        _DEFAULT[aug_name] = _one_update(aug_name)
    return _DEFAULT[aug_name]


def _compare(tag, got, want):
    """Per tensor at 1e-4.  One exception, with its own check: two correct fp32 evaluations of a conv layer may put an
    activation within rounding of zero on different sides of the ReLU, and ONE such element moves the conv weight
    gradients at and below that layer by ~1e-4 .. 1e-3 of their size (tests/test_gpu_fullsize.py).  When the two runs'
    ReLU branches differ -- in at most 8 elements, all within 1e-5 of zero on both sides -- the conv gradients are held
    to 5e-3 instead."""
    bad = []
    (losses, grads, _, par), (losses0, grads0, _, par0) = got, want
    flips = 0
    for a, b in zip(par["acts"], par0["acts"]):
        differ = (a > 0) != (b > 0)
        k = int(differ.sum())
        if k:
            assert float(torch.maximum(a[differ].abs(), b[differ].abs()).max()) <= 1e-5, tag
        flips += k
    assert flips <= 8, (tag, flips)
    assert set(losses) == set(losses0) and set(grads) == set(grads0) == {"critic", "actor", "cpc"}
    for k in losses0:
        if "loss" in k or "entropy" in k:
            e = rel_err(np.float64(losses[k]), np.float64(losses0[k]))
            if not e <= RTOL:
                bad.append((tag, k, e))
    n = 0
    for phase in grads0:
        assert set(grads[phase]) == set(grads0[phase])
        for k, v in grads0[phase].items():
            e = rel_err(grads[phase][k], v)
            n += 1
            # (2.3e-3 seen at this batch size: the first layer's bf16 form -- the default -- against its f32 forms puts ONE
            # CURL-phase activation of conv1 on the other side of zero)
            tol = 5e-3 if (flips and ".convs." in k) else RTOL
            if not (np.isfinite(e) and e <= tol):
                bad.append((tag, phase, k, e))
    assert n == 24 + 11 + 13
    assert not bad, bad


# (options, environment, augmentation, C-ABI entry points that must / must not have been called)
SWITCHES = [
    ("conv1_u8=hybrid", {"conv1_u8": "hybrid"}, {}, "random_crop", (), ()),
    ("conv1_u8=band", {"conv1_u8": "band"}, {}, "random_crop", (), ()),
    ("conv1_u8=rw", {"conv1_u8": "rw"}, {}, "random_crop", (), ()),
    ("conv1_u8=rwb", {"conv1_u8": "rwb"}, {}, "random_crop", (), ()),
    ("conv1_f32=band", {"conv1_f32": "band"}, {}, "color_jiggle", (), ()),
    ("s1_fwd=f43", {"s1_fwd": "f43"}, {}, "random_crop", (), ()),
    ("s1_fwd=f23", {"s1_fwd": "f23"}, {}, "random_crop", (), ()),
    ("s1_fwd=f23 (wide rows)", {"s1_fwd": "f23"}, {}, "wide", (), ()),
    ("s1_fwd=b3 (forward and data gradient)", {"s1_fwd": "b3"}, {}, "random_crop", (), ()),
    ("s1_fwd=b3 (wide rows)", {"s1_fwd": "b3"}, {}, "wide", (), ()),
    ("wgrad1_u8=f32", {"wgrad1_u8": "f32"}, {}, "random_crop", (), ()),
    ("wgrad1_u8=b16", {"wgrad1_u8": "b16"}, {}, "random_crop", (), ()),
    ("s1_wgrad=x", {"s1_wgrad": "x"}, {}, "random_crop", (), ()),
    ("s1_wgrad=x (wide rows)", {"s1_wgrad": "x"}, {}, "wide", (), ()),
    ("bwd_split=0", {"bwd_split": "0"}, {}, "random_crop", (), ()),
    ("bwd_split=1", {"bwd_split": "1"}, {}, "random_crop", (), ()),
    ("gemm_tile=6464", {"gemm_tile": "6464"}, {}, "random_crop", (), ()),
    ("gemm_tile=6432", {"gemm_tile": "6432"}, {}, "random_crop", (), ()),
    ("gemm_tile=3232", {"gemm_tile": "3232"}, {}, "random_crop", (), ()),
    ("gemm_tile=12864 (bf16x3 wherever whole tiles fit)", {"gemm_tile": "12864"}, {}, "random_crop", (), ()),
    ("gemm_mfma=f32", {"gemm_mfma": "f32"}, {}, "random_crop", (), ()),
    ("linear_bwd=split", {"linear_bwd": "split"}, {}, "random_crop", (), ()),
    ("gemm_mfma=b3", {"gemm_mfma": "b3"}, {}, "random_crop", (), ()),
    ("CURLA_CURL_HEAD=unfused", {}, {"CURLA_CURL_HEAD": "unfused"}, "random_crop", ("curla_curl_ce",), ("curla_curl_head",)),
    ("CURLA_FC_FWD=gemm", {}, {"CURLA_FC_FWD": "gemm"}, "random_crop", ("curla_gemm_multi",), ("curla_fc_fwd_multi",)),
    ("CURLA_FOUR_Q=0", {}, {"CURLA_FOUR_Q": "0"}, "random_crop", (), ("curla_gemm_nested",)),
    ("CURLA_STAGE_COPY=1", {}, {"CURLA_STAGE_COPY": "1"}, "random_crop", (), ()),
    ("CURLA_TORCH_ADAM=1", {}, {"CURLA_TORCH_ADAM": "1"}, "random_crop", (), ("curla_adam_step", "curla_adam_step_lerp")),
]


def test_default_path_calls_what_the_switches_replace():
    _, _, calls, _ = _default("random_crop")
    # (the fused Adam + target lerp launch is not in the list: this test's hooks replace optimizer.step, which makes the
    # agent take the plain step -- test_one_update_at_batch_256_... asserts it on an un-hooked agent)
    for name in ("curla_curl_head", "curla_fc_fwd_multi", "curla_gemm_nested", "curla_adam_step", "curla_linear_bwd",
                 "curla_conv3x3_s1_fwd_stack", "curla_conv3x3_s1_bwd_slabs", "curla_conv1_fwd2"):
        assert calls[name] >= 1, (name, dict(calls))
    assert calls["curla_curl_ce"] == 0 and calls["curla_gemm_multi"] == 0


@pytest.mark.parametrize("tag,options,env,aug_name,must_call,must_not_call", SWITCHES, ids=[s[0] for s in SWITCHES])
def test_whole_update_parity_under_switch(tag, options, env, aug_name, must_call, must_not_call):
    from curla_amd import _lib
    before = {k: _lib.get_option(k) for k in _lib.OPTIONS}
    got = _one_update(aug_name, options, env)
    assert {k: _lib.get_option(k) for k in _lib.OPTIONS} == before  # (the switch was scoped to the run)
    _compare(tag, got, _default(aug_name))
    calls = got[2]
    for name in must_call:
        assert calls[name] >= 1, (tag, name, dict(calls))
    for name in must_not_call:
        assert calls[name] == 0, (tag, name, dict(calls))


F32_FORMS = {"s1_fwd": "f23", "gemm_mfma": "f32", "conv1_u8": "rw", "wgrad1_u8": "f32"}  # every product on the exact f32-input MFMA


@pytest.mark.parametrize("forms", ["f32 forms (tight)", "default forms"])
def test_torch_adam_and_flat_adam_take_the_same_steps(forms):
    """CURLA_TORCH_ADAM=1 (torch's fused multi-tensor Adam) against the default FlatAdam over an even and an odd update
    with train.py's learning rates.  A first Adam step moves every element by lr * g / |g|, so single elements whose
    gradient is zero to rounding may differ by 2 lr per step; everything else agrees.
    Two configurations (the advisor's round-5 finding: the bound had been widened from 0.5 % to 10 % of the elements
    for the bf16x3 defaults, a window that could hide an optimizer regression): with every kernel pinned to its f32 form
    (``F32_FORMS``) the two optimizers see bit-identical gradients at the first step and the round-4 bound of 0.5 %
    holds; the default forms keep the wide bound, which there only absorbs one conv ReLU branch that lands on the other
    side of zero in the second update."""
    tight = forms.startswith("f32")
    opts = F32_FORMS if tight else {}
    flat = _one_update("random_crop", options=opts, lr=1e-3, steps=2)[3]
    fused = _one_update("random_crop", options=opts, env={"CURLA_TORCH_ADAM": "1"}, lr=1e-3, steps=2)[3]
    for k in ("critic", "actor", "target"):
        d = (flat[k] - fused[k]).abs()
        assert float(d.max()) <= 4 * 2 * 1e-3 + 1e-6, (k, float(d.max()))  # (a conv weight takes 4 steps in two updates)
        # (elements whose gradient is a sum of cancelling terms follow ONE conv ReLU branch that differs between the two
        # runs' second updates -- their first Adam steps differ in the last bit -- by more than 1 % of lr: a twentieth of
        # the critic in the worst run seen on the default forms; a different optimizer would move all of them)
        off = float((d > 1e-5).float().mean())
        assert off <= (0.005 if tight else 0.10), (forms, k, off)
        assert float(d.median()) <= 1e-6, (k, float(d.median()))
