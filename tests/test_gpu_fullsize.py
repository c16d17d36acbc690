"""BASELINE.json's configurations at their FULL sizes, compared with the oracle directly.

One even-step ``agent.update(replay_buffer, L, 0)`` -- critic phase, target soft update, actor / alpha phase, CURL
phase -- per configuration, on the MI355X, against the oracle's phases evaluated on the CPU from the same weights,
the same minibatch (the bytes the ring hands the kernels) and the same policy noise:

  c2  configs[1]: B = 512, 84x84x9 uint8 ring -> random_crop 76x76, 4 conv layers, hidden 1024, CURL + critic + actor
  c3  configs[2]: B = 512, 84x84x9 un-cropped, pixel_sac (no CURL phase)
  c5  configs[4] per GPU: B = 1024, 168x168x12, 6 conv layers, colour-jittered float observations -- the oracle is
      given the identical post-augmentation tensors (the jitter arithmetic itself is kornia's: parity unpinned)

ReLU branches.  A conv weight gradient at these sizes is a sum of ~600k (c2) to ~7M (c5) signed, largely cancelling
terms, so ONE activation whose pre-activation is within fp32 rounding of 0 -- positive in one evaluation of the network,
not in the other; 0-2 elements out of 20 million per layer -- moves it by 1e-4 .. 1e-3 of its size (the reference does
the same to itself: its own fp32 and fp64 evaluations differ by 2e-4 .. 3e-4 on these tensors).  The same holds
for the hidden units of the 1024-wide MLPs: one unit of one sample flipping moves a row of a trunk weight gradient by
~1/sqrt(B) of its size, and through d(loss)/dz everything below it.  The gradients are therefore compared with the
oracle evaluated under the SAME branch decisions (the derivative of each ReLU -- conv layers, Q trunks, actor trunk --
takes its branch from the device's activations: ``relu_branches`` / ``q_branches`` / ``trunk_branches`` in
oracle/curla_oracle.py; values are untouched), and the test asserts separately that the two sides disagree on at most a
few conv branches per layer, all at activations within 1e-5 of 0.  The un-aligned errors are written to the report as
well ("raw").

All learning rates are zero, so the four Adam steps inside the update leave the parameters where they were and every
phase of both sides is evaluated at identical weights (multi-step parameter trajectories are chaotic under fp32
reassociation, SURVEY.md D11 -- they are not what 1e-4 is about); the soft update runs with train.py's rates on both
sides.  Compared, each per tensor with max|a-b| / max|b| <= 1e-4 (curl_sac.py:349-423): the logged losses, z_a, z_pos
and the logits of the CURL phase, and EVERY gradient an optimizer consumes -- 24 critic tensors, 10 actor tensors,
log_alpha, 12 encoder tensors + W of the CURL phase."""
import os
import time

import numpy as np
import pytest
import torch

from tests._util import RTOL, rel_err
from tests.test_gpu_agent import HP, NullLogger, _copy_agent_into_oracle, grads_of

pytestmark = pytest.mark.gpu

REPORT = []


@pytest.fixture(scope="module", autouse=True)
def _report():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fullsize_parity.txt", "a") as f:
        for n, e in REPORT:
            f.write(f"{n:78s} {e:.3e}\n")


def check(name, got, ref, tol=RTOL):
    e = rel_err(got, ref)
    REPORT.append((name, e))
    return (name, e, tol) if not (np.isfinite(e) and e <= tol) else None


def _move_off_init(agent, oracle, layers, seed=5):
    """Identical, seeded perturbation of both sides: all nine taps of every conv live (the init is delta-orthogonal:
    centre tap only), biases and LayerNorm parameters away from 0 / 1, the target different from the online net."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in oracle.critic.items():
            scale = 0.05 if ".convs." in k else 0.02 if (k.endswith("bias") or ".ln." in k) else 0.0
            if scale:
                d = scale * torch.randn(v.shape, generator=g)
                v.add_(d)
                oracle.critic_target[k].add_(0.5 * d)
        for k, v in oracle.critic_target.items():
            if k.endswith("weight") and ".convs." not in k and ".ln." not in k:
                v.mul_(0.97)
        for k, v in oracle.actor.items():
            if k.endswith("bias") or ".ln." in k:
                v.add_(0.02 * torch.randn(v.shape, generator=g))
        agent.critic.load_state_dict({k: v.detach().clone() for k, v in oracle.critic.items()})
        agent.critic_target.load_state_dict({k: v.detach().clone() for k, v in oracle.critic_target.items()})
        convs = {k: v for k, v in oracle.critic.items() if ".convs." in k}
        agent.actor.load_state_dict({**{k: v.detach().clone() for k, v in oracle.actor.items()},
                                     **{k: v.detach().clone() for k, v in convs.items()}})


def _run(tag, obs_shape, in_hw, aug_name, layers, B, pixel_sac, capacity):
    import curla_amd
    from curla_amd import _lib
    from oracle import curla_oracle as O
    torch.manual_seed(31)
    np.random.seed(31)
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # (the oracle's CPU convs get slower beyond a few dozen threads)
    dev = torch.device("cuda")
    C = obs_shape[0]
    out_hw = tuple(obs_shape[1:])
    if aug_name == "random_crop":
        aug = curla_amd.RandomCrop(in_hw, out_hw)
    elif aug_name == "identity":
        aug = curla_amd.IdentityAugmentation(in_hw)
    else:
        aug = curla_amd.ColorJiggle(in_hw)
    hp = {**HP, "num_layers": layers, "alpha_lr": 0.0, "actor_lr": 0.0, "critic_lr": 0.0, "encoder_lr": 0.0}
    agent = curla_amd.CurlSacAgent(obs_shape, (2,), dev, aug, hidden_dim=1024, pixel_sac=pixel_sac, **hp)
    oracle = O.OracleAgent(obs_shape, (2,), hidden_dim=1024, pixel_sac=pixel_sac,
                           **{k: v for k, v in hp.items() if k != "log_interval"})
    _copy_agent_into_oracle(agent, oracle)
    _move_off_init(agent, oracle, layers)

    # the ring: uniform random bytes (BASELINE's synthetic data), filled on the device; a host copy feeds the oracle
    rb = curla_amd.ReplayBuffer((C,) + tuple(in_hw), (2,), capacity, B, dev, aug)
    g = torch.Generator(device=dev).manual_seed(7)
    for ring in (rb._obs_store, rb._next_store):
        ring[:] = torch.randint(0, 256, (ring.numel(),), dtype=torch.uint8, device=dev, generator=g)
    rb.actions.uniform_(-1, 1, generator=g)
    rb.rewards.normal_(generator=g)
    rb.not_dones.fill_(1.0)
    rb.not_dones[9::10] = 0.0
    rb.idx, rb.full = 0, True

    obs, act, rew, nxt, nd, kw = rb.sample_cpc_refs()
    pos = kw["obs_pos"]
    if aug_name == "color_jiggle":
        # float NHWC tensors, already augmented on the device: the oracle gets the same values in NCHW
        to_cpu = lambda r: r.src.permute(0, 3, 1, 2).contiguous().cpu()  # noqa: E731
        o_obs, o_nxt, o_pos = to_cpu(obs), to_cpu(nxt), to_cpu(pos)
        assert float(o_obs.min()) >= 0.0 and float(o_obs.max()) <= 255.001 and not torch.equal(o_obs, o_pos)
    else:
        # uint8 ring + indices + crop offsets: the oracle crops the same frames on the host (augmentations.py:65-73)
        idx = obs.idx.cpu().numpy()
        frames = rb._both if rb._both is not None else None
        assert frames is not None

        def host(ref):
            f = frames[ref.idx].cpu().numpy().transpose(0, 3, 1, 2)  # [B, C, H, W] uint8 (obs or next_obs half)
            if aug_name == "random_crop":
                f = O.random_crop(f, ref.h1.cpu().numpy(), ref.w1.cpu().numpy(), out_hw)
            return torch.from_numpy(np.ascontiguousarray(f)).float()
        o_obs, o_nxt, o_pos = host(obs), host(nxt), host(pos)
        assert np.array_equal(nxt.idx.cpu().numpy(), idx + capacity)
        if aug_name == "random_crop":
            assert not torch.equal(o_obs, o_pos)
    o_act, o_rew, o_nd = act.cpu().clone(), rew.cpu().clone(), nd.cpu().clone()
    noise_c, noise_a = torch.randn(B, 2), torch.randn(B, 2)

    # ---- oracle: the phases of OracleAgent.update() at step 0, learning rates zero (curl_sac.py:426-451)
    t0 = time.perf_counter()
    kwc = dict(num_layers=layers, log_std_min=-10, log_std_max=2)
    rc = O.critic_phase(oracle.actor, oracle.critic, oracle.critic_target, oracle.log_alpha, o_obs, o_act, o_rew, o_nxt,
                        o_nd, noise_c, discount=0.99, **kwc)
    ra = O.actor_phase(oracle.actor, oracle.critic, oracle.log_alpha, o_obs, noise_a,
                       target_entropy=oracle.target_entropy, **kwc)
    saved_target = {k: v.detach().clone() for k, v in oracle.critic_target.items()}  # (for the second critic pass)
    with torch.no_grad():
        for prefix, tau in (("Q1.", hp["critic_tau"]), ("Q2.", hp["critic_tau"]), ("encoder.", hp["encoder_tau"])):
            O.soft_update(oracle.critic, oracle.critic_target, tau, prefix)
    rp = None if pixel_sac else O.cpc_phase(oracle.critic, oracle.critic_target, oracle.W, o_obs, o_pos, num_layers=layers)
    t_oracle = time.perf_counter() - t0

    # ---- the device: ONE update() call on that minibatch and that noise
    rb.sample_cpc_refs = lambda: (obs, act, rew, nxt, nd, kw)
    noises = iter([noise_c.to(dev), noise_a.to(dev)])
    agent._noise = lambda ws, noise: (ws.noise.copy_(next(noises)), None)  # (buffer, rng): explicit noise, no in-kernel draw
    captured = {}
    real_step = agent.critic_optimizer.step

    def hidden(*ts):  # post-ReLU hidden activations of the MLPs -> the branches their derivatives took
        return [(t > 0).cpu() for t in ts]

    def critic_step():
        captured["critic"] = grads_of(agent.critic)
        w_ = agent._ws(B)
        captured["critic_q"] = hidden(w_.q_h1[0], w_.q_h2[0], w_.q_h1[1], w_.q_h2[1])
        real_step()
    agent.critic_optimizer.step = critic_step
    real_actor_step = agent.actor_optimizer.step

    def actor_step():
        w_ = agent._ws(B)
        captured["actor_mlp"] = hidden(w_.a_h1, w_.a_h2, w_.q_h1[0], w_.q_h2[0], w_.q_h1[1], w_.q_h2[1])
        real_actor_step()
    agent.actor_optimizer.step = actor_step
    before = agent._critic_flat.clone()
    L = NullLogger()
    agent.update(rb, L, 0)
    torch.cuda.synchronize()
    assert _lib._lib is not None
    assert torch.equal(before, agent._critic_flat), "learning rate 0: the parameters must not have moved"
    ws = agent._ws(B)

    bad = []
    bad.append(check(f"{tag} critic loss", L.scalars["train_critic/loss"], rc["loss"]))
    bad.append(check(f"{tag} actor loss", L.scalars["train_actor/loss"], ra["actor_loss"]))
    bad.append(check(f"{tag} alpha loss", L.scalars["train_alpha/loss"], ra["alpha_loss"]))
    bad.append(check(f"{tag} entropy", L.scalars["train_actor/entropy"], ra["entropy"]))
    assert len(captured["critic"]) == 8 + 2 * layers + 8 == len(rc["grads"])
    # ---- ReLU branches: the device's activations of obs under the (unmoved) online weights are still in the workspace
    # (the actor phase recomputed them from the same weights)
    branches, n_differ = [], 0
    for i in range(layers):
        dev_act = ws.acts_main[i].permute(0, 3, 1, 2).cpu()          # NHWC -> NCHW
        ref_act = rc["enc"][f"conv{i + 1}"]
        assert dev_act.shape == ref_act.shape
        bad.append(check(f"{tag} activations conv{i + 1}", dev_act, ref_act))
        pos = dev_act > 0
        differ = pos != (ref_act > 0)
        k = int(differ.sum())
        n_differ += k
        REPORT.append((f"{tag} conv{i + 1}: ReLU branches that differ (of {differ.numel()})", float(k)))
        assert k <= 4 + 2e-6 * differ.numel(), (i, k)
        if k:  # only where both sides are within rounding of zero
            assert float(torch.maximum(dev_act[differ].abs(), ref_act[differ].abs()).max()) <= 1e-5
        branches.append(pos)
        del dev_act, differ
    cq = captured["critic_q"]
    rcb = O.critic_phase(oracle.actor, oracle.critic, saved_target, oracle.log_alpha, o_obs, o_act, o_rew, o_nxt, o_nd,
                         noise_c, discount=0.99, relu_branches=branches, q_branches=[cq[0:2], cq[2:4]], **kwc)
    assert float((rcb["loss"] - rc["loss"]).abs()) == 0.0  # values are untouched, only derivative branches
    for k, v in rcb["grads"].items():
        check(f"{tag} critic grad {k} (raw: own branches on both sides)", captured["critic"][k], rc["grads"][k])
        bad.append(check(f"{tag} critic grad {k}", captured["critic"][k], v))
    actor_grads = {k: v for k, v in grads_of(agent.actor).items() if ".convs." not in k}
    assert len(actor_grads) == 10 == len(ra["grads"])
    am = captured["actor_mlp"]
    rab = O.actor_phase(oracle.actor, oracle.critic, oracle.log_alpha, o_obs, noise_a, target_entropy=oracle.target_entropy,
                        trunk_branches=am[0:2], q_branches=[am[2:4], am[4:6]], **kwc)
    assert float((rab["actor_loss"] - ra["actor_loss"]).abs()) == 0.0
    n_mlp = 0
    for nm, dev_b, ref_h in (("actor trunk", am[0:2], None), ("critic Q", cq, None)):
        n_mlp += sum(int(b.numel()) for b in dev_b)
    REPORT.append((f"{tag} MLP hidden units whose ReLU branch is taken from the device", float(n_mlp)))
    for k, v in rab["grads"].items():
        check(f"{tag} actor grad {k} (raw: own branches on both sides)", actor_grads[k], ra["grads"][k])
        bad.append(check(f"{tag} actor grad {k}", actor_grads[k], v))
    bad.append(check(f"{tag} log_alpha grad", agent.log_alpha.grad.detach().cpu().reshape(1),
                     rab["log_alpha_grad"].reshape(1)))
    # the target after the soft update (utils.py:37-41)
    tsd = agent.critic_target.state_dict()
    for k in ("encoder.convs.0.weight", f"encoder.convs.{layers - 1}.weight", "encoder.fc.weight", "Q1.trunk.2.weight"):
        bad.append(check(f"{tag} target after soft update {k}", tsd[k].cpu(), oracle.critic_target[k].detach(), 1e-6))
    if rp is not None:
        bad.append(check(f"{tag} curl loss", L.scalars["train/curl_loss"], rp["loss"]))
        bad.append(check(f"{tag} cpc z_a", ws.z_c.cpu(), rp["z_a"]))
        bad.append(check(f"{tag} cpc z_pos", ws.z_pos.cpu(), rp["z_pos"]))
        lg = ws.logits.cpu()
        bad.append(check(f"{tag} cpc logits (minus row max)", lg - lg.max(1, keepdim=True)[0], rp["logits"]))
        cpc = {k: v for k, v in grads_of(agent.critic).items() if k.startswith("encoder.")}
        assert len(cpc) == 4 + 2 * layers == len(rp["grads"])
        rpb = O.cpc_phase(oracle.critic, oracle.critic_target, oracle.W, o_obs, o_pos, num_layers=layers,
                          relu_branches=branches)
        for k, v in rpb["grads"].items():
            if ".convs." in k:
                check(f"{tag} cpc grad {k} (raw: own branches on both sides)", cpc[k], rp["grads"][k])
            bad.append(check(f"{tag} cpc grad {k}", cpc[k], v))
        bad.append(check(f"{tag} cpc grad W", agent.CURL.W.grad.detach().cpu(), rp["W_grad"]))
    REPORT.append((f"{tag} (oracle seconds on {torch.get_num_threads()} host threads, one pass)", t_oracle))
    REPORT.append((f"{tag} ReLU branches that differ, all layers", float(n_differ)))
    bad = [b for b in bad if b is not None]
    assert not bad, bad


def test_c2_full_size_update_vs_oracle():
    _run("c2 B=512 84->76 L=4", (9, 76, 76), (84, 84), "random_crop", 4, 512, False, capacity=2048)


def test_c3_full_size_pixel_sac_update_vs_oracle():
    _run("c3 B=512 84x84 pixel_sac", (9, 84, 84), (84, 84), "identity", 4, 512, True, capacity=2048)


def test_c5_full_size_update_vs_oracle_on_identical_post_augmentation_tensors():
    _run("c5 B=1024 168x168x12 L=6", (12, 168, 168), (168, 168), "color_jiggle", 6, 1024, False, capacity=1024)
