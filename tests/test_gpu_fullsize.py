"""BASELINE.json's configurations at their FULL sizes, compared with the oracle directly, float64 as the arbiter.

One even-step ``agent.update(replay_buffer, L, 0)`` -- critic phase, target soft update, actor / alpha phase, CURL
phase -- per configuration, on the MI355X, against the oracle's phases evaluated on the CPU from the same weights,
the same minibatch (the bytes the ring hands the kernels) and the same policy noise:

  c2  configs[1]: B = 512, 84x84x9 uint8 ring -> random_crop 76x76, 4 conv layers, hidden 1024, CURL + critic + actor
  c3  configs[2]: B = 512, 84x84x9 un-cropped, pixel_sac (no CURL phase)
  c5  configs[4] per GPU: B = 1024, 168x168x12, 6 conv layers, colour-jittered float observations -- the oracle is
      given the identical post-augmentation tensors (the jitter arithmetic itself is kornia's: parity unpinned)

Three evaluations of every phase are compared: the DEVICE (fp32, HIP kernels), the oracle in fp32 (the reference's own
arithmetic: PyTorch CPU fp32, pinned by the golden vectors) and the oracle in FLOAT64 (same function, parameters / inputs
/ noise cast to double): the arbiter.  The float64 evaluation is PyTorch's, with its own float64 kernels on the device
(nothing of this library is involved; configs[4] takes 15 s that way instead of 240 s on the host cores), pinned to the
host's float64 evaluation of the configs[1] critic phase at 1e-10.

Values -- losses, conv activations, z_a, z_pos, logits -- are held to 1e-4 against the fp32 oracle and, per tensor, to
``e_hip <= max(1e-4, 2 e_ref)`` where e_hip = err(device, fp64) and e_ref = err(fp32 oracle, fp64).

Gradients need one more notion.  A conv weight gradient at these sizes is a sum of ~600k (c2) to ~7M (c5) signed,
largely cancelling terms, so ONE activation whose pre-activation is within fp32 rounding of 0 -- positive in one fp32
evaluation of the network, not in another -- moves it by 1e-4 .. 1e-3 of its size, and so does one hidden unit of the
1024-wide MLPs for the layers below it.  These events are DISCRETE and few (0-6 per layer at B = 512, ~60 per evaluation
at configs[4]), so the un-aligned error of any fp32 evaluation -- the reference's own included -- is a Poisson draw: in
this test's first run e_hip / e_ref ranged from 0.8 to 13 over the conv gradients and reached 9000 on an MLP tensor where
the device had one such unit and the reference none.  What IS a property of the kernels, and is asserted, per tensor:

  (A) the device is exact up to its branch decisions:  err(device, fp64 differentiated along the DEVICE's ReLU
      branches) <= 1e-4 for every gradient an optimizer consumes (the fp64 forward pass is differentiated again with
      each ReLU derivative taking its branch from the device's activations; values are untouched);
  (B) the device's branch decisions are legitimate: wherever the device's branch differs from float64's, the
      activation is within 1e-5 of zero on both sides; there are at most 2 n_ref + 8 such places per configuration,
      n_ref being the number of places where the fp32 ORACLE differs from float64;
  (C) the reference does the same to itself: (A) and (B) hold for the fp32 oracle in the device's place --
      err(fp32 oracle, fp64 along the fp32 oracle's branches) <= 1e-4, its disagreements with float64 all within
      1e-5 of zero -- so its un-aligned error e_ref (1.6e-4 .. 8e-4 on the conv gradients) is branch decisions too;
  (D) un-aligned, where the statistics carry it (the maximum over all tensors of a configuration):
      max e_hip <= max(1e-4, 4 max e_ref); and per tensor a hard cap, e_hip <= 5e-3 (round 6: what ONE near-zero ReLU can
      do to a tensor is ~1e-3 of it; a dropped term or tap is far above the cap).
Both un-aligned columns (e_hip, e_ref) and both aligned ones are written per tensor to gpurun_out/fullsize_arbiter.txt
(committed as profiles/r04_fullsize_parity.txt).  The device's branches are also counted against the fp32 oracle's,
conv layers and MLP hidden units alike (at most 64 per conv layer, 32 per MLP layer, all within 1e-5 of zero).

All learning rates are zero, so the four Adam steps inside the update leave the parameters where they were and every
phase of both sides is evaluated at identical weights (multi-step parameter trajectories are chaotic under fp32
reassociation, SURVEY.md D11 -- they are not what 1e-4 is about); the soft update runs with train.py's rates on both
sides.  Compared, each per tensor with max|a-b| / max|b| (curl_sac.py:349-423): the logged losses, z_a, z_pos and the
logits of the CURL phase, and EVERY gradient an optimizer consumes -- 24 critic tensors, 10 actor tensors, log_alpha,
12 encoder tensors + W of the CURL phase."""
import os
import time

import numpy as np
import pytest
import torch

from tests._util import RTOL, rel_err
from tests.test_gpu_agent import HP, NullLogger, _copy_agent_into_oracle, grads_of

pytestmark = pytest.mark.gpu

REPORT = []   # (name, number): values against the fp32 oracle, branch counts, timings
ARBITER = []  # (name, e_hip, e_ref, a_hip, a_ref): errors against the float64 evaluation (a_*: along own branches)
NEAR_ZERO = 1e-5
UNALIGNED_CAP = 5e-3  # per-tensor bound on a gradient's un-aligned error against float64 (grad_arbiter)


@pytest.fixture(scope="module", autouse=True)
def _report():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fullsize_parity.txt", "a") as f:
        for n, e in REPORT:
            f.write(f"{n:86s} {e:.3e}\n")
    with open("gpurun_out/fullsize_arbiter.txt", "a") as f:
        f.write("# errors against the oracle's FLOAT64 evaluation of the same phase (max|a-b| / max|b| per tensor)\n"
                "#   e_hip / e_ref: device / fp32 oracle against float64, NO branch alignment\n"
                "#   a_hip / a_ref: device / fp32 oracle against float64 differentiated along THEIR OWN ReLU branches\n"
                "# asserted: a_hip <= 1e-4, a_ref <= 1e-4 per tensor; values: e_hip <= max(1e-4, 2 e_ref) per tensor;\n"
                "#           per configuration max e_hip <= max(1e-4, 4 max e_ref)\n")
        f.write(f"# {'tensor':70s} {'e_hip':>10s} {'e_ref':>10s} {'a_hip':>10s} {'a_ref':>10s}\n")
        for n, eh, er, ah, ar in ARBITER:
            cols = " ".join(f"{v:10.3e}" if v is not None else f"{'-':>10s}" for v in (eh, er, ah, ar))
            f.write(f"{n:72s} {cols}\n")


def check(name, got, ref, tol=RTOL):
    e = rel_err(got, ref)
    REPORT.append((name, e))
    return (name, e, tol) if not (np.isfinite(e) and e <= tol) else None


def value_arbiter(name, got, ref32, ref64):
    """A value (loss, activation, feature, logit): e_hip = err(device, fp64), e_ref = err(fp32 oracle, fp64); fails
    when the device is further from the exact result than 1e-4 AND than twice the reference's own fp32 evaluation."""
    e_hip, e_ref = rel_err(got, ref64), rel_err(ref32, ref64)
    ARBITER.append((name, e_hip, e_ref, None, None))
    ok = np.isfinite(e_hip) and e_hip <= max(RTOL, 2.0 * e_ref)
    return None if ok else (name + " [fp64 arbiter]", e_hip, max(RTOL, 2.0 * e_ref))


def grad_arbiter(name, got, ref32, g64_own, g64_along_hip, g64_along_ref, worst):
    """A gradient: un-aligned errors reported (and folded into the configuration's maxima), aligned ones asserted:
    (A) the device against float64 along the device's branches, (C) the fp32 oracle against float64 along its own."""
    e_hip, e_ref = rel_err(got, g64_own), rel_err(ref32, g64_own)
    a_hip, a_ref = rel_err(got, g64_along_hip), rel_err(ref32, g64_along_ref)
    ARBITER.append((name, e_hip, e_ref, a_hip, a_ref))
    worst[0], worst[1] = max(worst[0], e_hip), max(worst[1], e_ref)
    bad = []
    # Per tensor, un-aligned: a hard cap.  The ratio form e_hip <= max(1e-4, 4 e_ref) is asserted per CONFIGURATION (end
    # of _run) and not per tensor because a single near-zero ReLU on one side and none on the other makes a tensor's
    # ratio anything (1900 seen on an MLP bias with e_hip = 1.6e-4: see there); what one such event CAN do to a tensor is
    # bounded, though -- ~1e-3 of it at these sizes (observed maximum over rounds 4-6: 1.6e-3) -- so every tensor is held
    # to 5e-3 un-aligned: a kernel that drops a term, a tap or a sample fails here whatever the branches do.
    if not (np.isfinite(e_hip) and e_hip <= UNALIGNED_CAP):
        bad.append((name + " [device vs fp64, un-aligned, per-tensor cap]", e_hip, UNALIGNED_CAP))
    if not (np.isfinite(a_hip) and a_hip <= RTOL):
        bad.append((name + " [device vs fp64 along the device's branches]", a_hip, RTOL))
    if not (np.isfinite(a_ref) and a_ref <= RTOL):
        bad.append((name + " [fp32 oracle vs fp64 along the fp32 oracle's branches]", a_ref, RTOL))
    return bad


def branch_flips(tag, what, a, b, limit=None):
    """Units whose ReLU branch differs between two evaluations (pre- or post-activation values, any float dtype):
    how many; every one of them must be within NEAR_ZERO of zero on both sides."""
    a, b = a.cpu(), b.cpu()
    differ = (a > 0) != (b > 0)
    k = int(differ.sum())
    worst = float(torch.maximum(a[differ].abs().double(), b[differ].abs().double()).max()) if k else 0.0
    REPORT.append((f"{tag} {what}: ReLU branches that differ (of {differ.numel()})", float(k)))
    if k:
        REPORT.append((f"{tag} {what}: largest |activation| at a differing branch", worst))
    assert worst <= NEAR_ZERO, (what, worst)
    if limit is not None:
        assert k <= limit, (what, k, limit)
    return k


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


def _make(obs_shape, in_hw, aug_name, layers, B, pixel_sac, capacity, **hp_extra):
    """(agent, oracle, ring, hp): an agent and the oracle on the same seeded, moved-off-init weights, all learning rates
    zero, and a ring of uniform random bytes (BASELINE's synthetic data) filled on the device."""
    import curla_amd
    from oracle import curla_oracle as O
    torch.manual_seed(31)
    np.random.seed(31)
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # (the oracle's CPU convs get no faster beyond 16 threads here)
    dev = torch.device("cuda")
    C = obs_shape[0]
    out_hw = tuple(obs_shape[1:])
    if aug_name == "random_crop":
        aug = curla_amd.RandomCrop(in_hw, out_hw)
    elif aug_name == "random_crop_default":
        # the reference's own arithmetic decides the crop: ceil(0.84 * side) (augmentations.py:21-24)
        aug = curla_amd.RandomCrop(in_hw)
        assert aug.output_shape == out_hw, (aug.output_shape, out_hw)
    elif aug_name == "identity":
        aug = curla_amd.IdentityAugmentation(in_hw)
    else:
        aug = curla_amd.ColorJiggle(in_hw)
    hp = {**HP, "num_layers": layers, "alpha_lr": 0.0, "actor_lr": 0.0, "critic_lr": 0.0, "encoder_lr": 0.0, **hp_extra}
    agent = curla_amd.CurlSacAgent(obs_shape, (2,), dev, aug, hidden_dim=1024, pixel_sac=pixel_sac, **hp)
    oracle = O.OracleAgent(obs_shape, (2,), hidden_dim=1024, pixel_sac=pixel_sac,
                           **{k: v for k, v in hp.items() if k != "log_interval"})
    _copy_agent_into_oracle(agent, oracle)
    _move_off_init(agent, oracle, layers)

    rb = curla_amd.ReplayBuffer((C,) + tuple(in_hw), (2,), capacity, B, dev, aug)
    g = torch.Generator(device=dev).manual_seed(7)
    for ring in (rb._obs_store, rb._next_store):
        ring[:] = torch.randint(0, 256, (ring.numel(),), dtype=torch.uint8, device=dev, generator=g)
    rb.actions.uniform_(-1, 1, generator=g)
    rb.rewards.normal_(generator=g)
    rb.not_dones.fill_(1.0)
    rb.not_dones[9::10] = 0.0
    rb.idx, rb.full = 0, True
    return agent, oracle, rb, hp


def _host_minibatch(rb, sample, aug_name, out_hw, capacity):
    """The minibatch the ring handed the kernels, as the float NCHW tensors the reference's sample_cpc() returns
    (utils.py:144-187): (obs, next_obs, pos)."""
    from oracle import curla_oracle as O
    obs, act, rew, nxt, nd, kw = sample
    pos = kw["obs_pos"]
    if aug_name == "color_jiggle":
        # float NHWC tensors, already augmented on the device: the oracle gets the same values in NCHW
        to_cpu = lambda r: r.src.permute(0, 3, 1, 2).contiguous().cpu()  # noqa: E731
        o_obs, o_nxt, o_pos = to_cpu(obs), to_cpu(nxt), to_cpu(pos)
        assert float(o_obs.min()) >= 0.0 and float(o_obs.max()) <= 255.001 and not torch.equal(o_obs, o_pos)
        return o_obs, o_nxt, o_pos
    # uint8 ring + indices + crop offsets: the oracle crops the same frames on the host (augmentations.py:65-73)
    idx = obs.idx.cpu().numpy()
    frames = rb._both if rb._both is not None else None
    assert frames is not None

    def host(ref):
        f = frames[ref.idx].cpu().numpy().transpose(0, 3, 1, 2)  # [B, C, H, W] uint8 (obs or next_obs half)
        if aug_name.startswith("random_crop"):
            f = O.random_crop(f, ref.h1.cpu().numpy(), ref.w1.cpu().numpy(), out_hw)
        return torch.from_numpy(np.ascontiguousarray(f)).float()
    o_obs, o_nxt, o_pos = host(obs), host(nxt), host(pos)
    assert np.array_equal(nxt.idx.cpu().numpy(), idx + capacity)
    if aug_name.startswith("random_crop"):
        assert not torch.equal(o_obs, o_pos)
    return o_obs, o_nxt, o_pos


def _run(tag, obs_shape, in_hw, aug_name, layers, B, pixel_sac, capacity):
    from curla_amd import _lib
    from oracle import curla_oracle as O
    agent, oracle, rb, hp = _make(obs_shape, in_hw, aug_name, layers, B, pixel_sac, capacity)
    dev = torch.device("cuda")
    out_hw = tuple(obs_shape[1:])

    obs, act, rew, nxt, nd, kw = rb.sample_cpc_refs()
    o_obs, o_nxt, o_pos = _host_minibatch(rb, (obs, act, rew, nxt, nd, kw), aug_name, out_hw, capacity)
    o_act, o_rew, o_nd = act.cpu().clone(), rew.cpu().clone(), nd.cpu().clone()
    noise_c, noise_a = torch.randn(B, 2), torch.randn(B, 2)

    # ---- the device: ONE update() call on that minibatch and that noise
    rb.sample_cpc_refs = lambda: (obs, act, rew, nxt, nd, kw)
    noises = iter([noise_c.to(dev), noise_a.to(dev)])
    agent._noise = lambda ws, noise: (ws.noise.copy_(next(noises)), None)  # (buffer, rng): explicit noise, no in-kernel draw
    captured = {}
    real_step = agent.critic_optimizer.step

    def critic_step():  # (post-ReLU hidden activations of the twin-Q MLPs: their signs are the device's branches)
        captured["critic"] = grads_of(agent.critic)
        w_ = agent._ws(B)
        captured["critic_qh"] = [t.detach().cpu().clone() for t in (w_.q_h1[0], w_.q_h2[0], w_.q_h1[1], w_.q_h2[1])]
        real_step()
    agent.critic_optimizer.step = critic_step
    real_actor_step = agent.actor_optimizer.step

    def actor_step():
        w_ = agent._ws(B)
        hs = (w_.a_h1, w_.a_h2, w_.q_h1[0], w_.q_h2[0], w_.q_h1[1], w_.q_h2[1])
        captured["actor_mlph"] = [t.detach().cpu().clone() for t in hs]
        real_actor_step()
    agent.actor_optimizer.step = actor_step
    before = agent._critic_flat.clone()
    L = NullLogger()
    agent.update(rb, L, 0)
    torch.cuda.synchronize()
    assert _lib._lib is not None
    assert torch.equal(before, agent._critic_flat), "learning rate 0: the parameters must not have moved"
    ws = agent._ws(B)
    # the device's activations of obs under the (unmoved) online weights are still in the workspace (the actor phase
    # recomputed them from the same weights; the CURL phase of an even step reuses them)
    dev_acts = [ws.acts_main[i].permute(0, 3, 1, 2).cpu() for i in range(layers)]  # NHWC -> NCHW
    hip_conv = [a > 0 for a in dev_acts]
    cqh, amh = captured["critic_qh"], captured["actor_mlph"]
    pos_of = lambda ts: [t > 0 for t in ts]  # noqa: E731

    # ---- the fp32 oracle: the phases of OracleAgent.update() at step 0, learning rates zero (curl_sac.py:426-451)
    t0 = time.perf_counter()
    kwc = dict(num_layers=layers, log_std_min=-10, log_std_max=2)
    rc = O.critic_phase(oracle.actor, oracle.critic, oracle.critic_target, oracle.log_alpha, o_obs, o_act, o_rew, o_nxt,
                        o_nd, noise_c, discount=0.99, **kwc)
    ra = O.actor_phase(oracle.actor, oracle.critic, oracle.log_alpha, o_obs, noise_a,
                       target_entropy=oracle.target_entropy, **kwc)
    # float64: the oracle's own functions, evaluated by PyTorch's float64 kernels ON THE DEVICE (nothing of this
    # library is involved; 15 s instead of 240 s for configs[4]) -- and pinned to the host's float64 evaluation below
    d = lambda x: O.as_dtype(x, torch.float64, dev)  # noqa: E731
    on_dev = lambda masks: [m.to(dev) for m in masks]  # noqa: E731
    target64 = d(oracle.critic_target)  # (before the soft update)
    with torch.no_grad():
        for prefix, tau in (("Q1.", hp["critic_tau"]), ("Q2.", hp["critic_tau"]), ("encoder.", hp["encoder_tau"])):
            O.soft_update(oracle.critic, oracle.critic_target, tau, prefix)
    rp = None if pixel_sac else O.cpc_phase(oracle.critic, oracle.critic_target, oracle.W, o_obs, o_pos, num_layers=layers)
    t_oracle = time.perf_counter() - t0
    ref_conv = [rc["enc"][f"conv{i + 1}"] > 0 for i in range(layers)]

    bad, worst = [], [0.0, 0.0]   # worst: the configuration's largest un-aligned gradient errors (device, fp32 oracle)
    n_hip64 = n_ref64 = 0         # branch disagreements with float64, all ReLUs of the configuration

    # ---- critic phase (curl_sac.py:349-371): float64, differentiated three times from one forward pass
    t0 = time.perf_counter()
    d_obs = d(o_obs)
    la64 = d(oracle.log_alpha)
    rc64 = O.critic_phase(d(oracle.actor), d(oracle.critic), target64, la64, d_obs, d(o_act), d(o_rew),
                          d(o_nxt), d(o_nd), d(noise_c), discount=0.99, regrad=True, **kwc)
    g64 = rc64["grads"]
    g64_hip = rc64["regrad"](relu_branches=on_dev(hip_conv),
                             q_branches=[on_dev(pos_of(cqh[0:2])), on_dev(pos_of(cqh[2:4]))])
    g64_ref = rc64["regrad"](relu_branches=on_dev(ref_conv), q_branches=[on_dev(pos_of(rc["q_hidden"][0:2])),
                                                                          on_dev(pos_of(rc["q_hidden"][2:4]))])
    del rc64["regrad"]
    if B <= 512 and not pixel_sac:
        # the device-side float64 evaluation IS the host's: the same phase in float64 on the CPU (configs[1] only: it
        # takes 7 s there, 70 s at configs[4]), every gradient to 1e-10
        h = lambda x: O.as_dtype(x, torch.float64, "cpu")  # noqa: E731
        rc64h = O.critic_phase(h(oracle.actor), h(oracle.critic), h(target64), h(oracle.log_alpha), h(o_obs), h(o_act),
                               h(o_rew), h(o_nxt), h(o_nd), h(noise_c), discount=0.99, **kwc)
        worst64 = max(rel_err(g64[k], v) for k, v in rc64h["grads"].items())
        REPORT.append((f"{tag} float64 critic gradients, device-side torch vs host torch: worst", worst64))
        assert worst64 <= 1e-10 and rel_err(rc64["loss"], rc64h["loss"]) <= 1e-12
        del rc64h
    t_oracle64 = time.perf_counter() - t0
    bad.append(check(f"{tag} critic loss", L.scalars["train_critic/loss"], rc["loss"]))
    bad.append(value_arbiter(f"{tag} critic loss", L.scalars["train_critic/loss"], rc["loss"], rc64["loss"]))
    n_differ = 0
    for i in range(layers):
        name = f"conv{i + 1}"
        a32, a64 = rc["enc"][name], rc64["enc"][name]
        assert dev_acts[i].shape == a32.shape
        bad.append(check(f"{tag} activations {name}", dev_acts[i], a32))
        bad.append(value_arbiter(f"{tag} activations {name}", dev_acts[i], a32, a64))
        # few, and only where both sides are within rounding of zero (observed: <= 6 per layer at B = 512, <= 30 at
        # configs[4]'s 2e8 activations per layer)
        n_differ += branch_flips(tag, f"{name}, device vs fp32 oracle", dev_acts[i], a32, 64)
        n_hip64 += branch_flips(tag, f"{name}, device vs float64", dev_acts[i], a64)
        n_ref64 += branch_flips(tag, f"{name}, fp32 oracle vs float64", a32, a64)
        rc64["enc"][name] = None
    n_mlp = 0
    for j, nm in enumerate(("Q1 hidden 1", "Q1 hidden 2", "Q2 hidden 1", "Q2 hidden 2")):
        n_mlp += branch_flips(tag, f"critic phase {nm}, device vs fp32 oracle", cqh[j], rc["q_hidden"][j], 32)
        n_hip64 += branch_flips(tag, f"critic phase {nm}, device vs float64", cqh[j], rc64["q_hidden"][j])
        n_ref64 += branch_flips(tag, f"critic phase {nm}, fp32 oracle vs float64", rc["q_hidden"][j], rc64["q_hidden"][j])
    assert len(captured["critic"]) == 8 + 2 * layers + 8 == len(rc["grads"])
    for k in rc["grads"]:
        bad += grad_arbiter(f"{tag} critic grad {k}", captured["critic"][k], rc["grads"][k], g64[k], g64_hip[k],
                            g64_ref[k], worst)
    del rc64, g64, g64_hip, g64_ref

    # ---- actor / alpha phase (curl_sac.py:373-404): no conv gradients; the branches are the MLPs' hidden units
    t0 = time.perf_counter()
    ra64 = O.actor_phase(d(oracle.actor), d(oracle.critic), la64, d_obs, d(noise_a),
                         target_entropy=oracle.target_entropy, regrad=True, **kwc)
    g64 = ra64["grads"]
    g64_hip = ra64["regrad"](trunk_branches=on_dev(pos_of(amh[0:2])),
                             q_branches=[on_dev(pos_of(amh[2:4])), on_dev(pos_of(amh[4:6]))])
    ref_h = ra["trunk_hidden"] + ra["q_hidden"]
    g64_ref = ra64["regrad"](trunk_branches=on_dev(pos_of(ref_h[0:2])),
                             q_branches=[on_dev(pos_of(ref_h[2:4])), on_dev(pos_of(ref_h[4:6]))])
    del ra64["regrad"]
    t_oracle64 += time.perf_counter() - t0
    for nm, key, rk in (("actor loss", "train_actor/loss", "actor_loss"), ("alpha loss", "train_alpha/loss", "alpha_loss"),
                        ("entropy", "train_actor/entropy", "entropy")):
        bad.append(check(f"{tag} {nm}", L.scalars[key], ra[rk]))
        bad.append(value_arbiter(f"{tag} {nm}", L.scalars[key], ra[rk], ra64[rk]))
    h64 = ra64["trunk_hidden"] + ra64["q_hidden"]
    for j, nm in enumerate(("trunk hidden 1", "trunk hidden 2", "Q1 hidden 1", "Q1 hidden 2", "Q2 hidden 1", "Q2 hidden 2")):
        n_mlp += branch_flips(tag, f"actor phase {nm}, device vs fp32 oracle", amh[j], ref_h[j], 32)
        n_hip64 += branch_flips(tag, f"actor phase {nm}, device vs float64", amh[j], h64[j])
        n_ref64 += branch_flips(tag, f"actor phase {nm}, fp32 oracle vs float64", ref_h[j], h64[j])
    REPORT.append((f"{tag} MLP hidden units whose ReLU branch differs, device vs fp32 oracle (of "
                   f"{sum(int(t.numel()) for t in cqh + amh)})", float(n_mlp)))
    actor_grads = {k: v for k, v in grads_of(agent.actor).items() if ".convs." not in k}
    assert len(actor_grads) == 10 == len(ra["grads"])
    for k in ra["grads"]:
        bad += grad_arbiter(f"{tag} actor grad {k}", actor_grads[k], ra["grads"][k], g64[k], g64_hip[k], g64_ref[k], worst)
    la_grad = agent.log_alpha.grad.detach().cpu().reshape(1)
    bad.append(check(f"{tag} log_alpha grad", la_grad, ra["log_alpha_grad"].reshape(1)))
    bad.append(value_arbiter(f"{tag} log_alpha grad", la_grad, ra["log_alpha_grad"].reshape(1),
                             ra64["log_alpha_grad"].reshape(1)))
    del ra64, g64, g64_hip, g64_ref

    # ---- the target after the soft update (utils.py:37-41)
    tsd = agent.critic_target.state_dict()
    for k in ("encoder.convs.0.weight", f"encoder.convs.{layers - 1}.weight", "encoder.fc.weight", "Q1.trunk.2.weight"):
        bad.append(check(f"{tag} target after soft update {k}", tsd[k].cpu(), oracle.critic_target[k].detach(), 1e-6))

    # ---- CURL phase (curl_sac.py:406-423)
    if rp is not None:
        t0 = time.perf_counter()
        rp64 = O.cpc_phase(d(oracle.critic), d(oracle.critic_target), d(oracle.W), d_obs, d(o_pos), num_layers=layers,
                           regrad=True)
        g64, w64 = rp64["grads"], rp64["W_grad"]
        g64_hip, w64_hip = rp64["regrad"](relu_branches=on_dev(hip_conv))
        g64_ref, w64_ref = rp64["regrad"](relu_branches=on_dev([rp["enc"][f"conv{i + 1}"] > 0 for i in range(layers)]))
        del rp64["regrad"], rp64["enc"]
        t_oracle64 += time.perf_counter() - t0
        lg = ws.logits.cpu()
        lg = lg - lg.max(1, keepdim=True)[0]
        for nm, got, key in (("curl loss", L.scalars["train/curl_loss"], "loss"), ("cpc z_a", ws.z_c.cpu(), "z_a"),
                             ("cpc z_pos", ws.z_pos.cpu(), "z_pos"), ("cpc logits (minus row max)", lg, "logits")):
            bad.append(check(f"{tag} {nm}", got, rp[key]))
            bad.append(value_arbiter(f"{tag} {nm}", got, rp[key], rp64[key]))
        cpc = {k: v for k, v in grads_of(agent.critic).items() if k.startswith("encoder.")}
        assert len(cpc) == 4 + 2 * layers == len(rp["grads"])
        for k in rp["grads"]:
            bad += grad_arbiter(f"{tag} cpc grad {k}", cpc[k], rp["grads"][k], g64[k], g64_hip[k], g64_ref[k], worst)
        bad += grad_arbiter(f"{tag} cpc grad W", agent.CURL.W.grad.detach().cpu(), rp["W_grad"], w64, w64_hip, w64_ref, worst)
        del rp64, g64, g64_hip, g64_ref
    del d_obs

    # ---- (B) the device disagrees with float64 about as often as the reference's own fp32 evaluation does,
    #      (D) and its worst un-aligned error is of the size of the reference's own
    REPORT.append((f"{tag} ReLU branches that differ from float64's, all ReLUs: device", float(n_hip64)))
    REPORT.append((f"{tag} ReLU branches that differ from float64's, all ReLUs: fp32 oracle", float(n_ref64)))
    REPORT.append((f"{tag} conv ReLU branches that differ, device vs fp32 oracle, all layers", float(n_differ)))
    REPORT.append((f"{tag} largest un-aligned gradient error against float64: device", worst[0]))
    REPORT.append((f"{tag} largest un-aligned gradient error against float64: fp32 oracle", worst[1]))
    REPORT.append((f"{tag} (fp32 oracle seconds on {torch.get_num_threads()} host threads)", t_oracle))
    REPORT.append((f"{tag} (float64 arbiter seconds, PyTorch float64 on the device)", t_oracle64))
    # (round 5: with the stride-1 convs on the bf16x3 form the device disagrees with float64 as often as the fp32 oracle
    # does -- 108 against 109 branches at configs[4], 7 / 5 and 16 / 13 at configs[1] / [2]; F(4,3) had 154 -- so the
    # bound is 1.5 x the reference's own count, not 2 x)
    assert n_hip64 <= 1.5 * n_ref64 + 8, (n_hip64, n_ref64)
    # Why the un-aligned GRADIENT errors are bounded per configuration and not per tensor (per tensor they are
    # reported, second column of gpurun_out/fullsize_arbiter.txt): the events are discrete and few.  One hidden unit of
    # a 1024-wide MLP layer, or one conv activation, whose pre-activation is within fp32 rounding of zero moves the
    # gradients below it by 1e-4 .. 1e-3 of their size, on whichever side it happens: in the round-5 run
    # `critic grad Q1.trunk.0.bias` at configs[4] is 1.6e-4 from float64 on the device and 8.6e-8 for the fp32 oracle
    # (ratio 1900: the device has one such unit in Q1's first hidden layer, the oracle none), while
    # `critic grad encoder.convs.0.weight` at configs[1] is 2.6e-4 / 2.4e-4.  A per-tensor ratio test would fail or pass
    # by the draw; aligned along either side's own branches every tensor is within 5e-6 (asserted above, per tensor).
    if not worst[0] <= max(RTOL, 4.0 * worst[1]):
        bad.append((f"{tag} largest un-aligned gradient error [device vs 4 x fp32 oracle]", worst[0], 4.0 * worst[1]))
    bad = [b for b in bad if b is not None]
    assert not bad, bad


def test_c2_full_size_update_vs_oracle():
    _run("c2 B=512 84->76 L=4", (9, 76, 76), (84, 84), "random_crop", 4, 512, False, capacity=2048)


def test_c3_full_size_pixel_sac_update_vs_oracle():
    _run("c3 B=512 84x84 pixel_sac", (9, 84, 84), (84, 84), "identity", 4, 512, True, capacity=2048)


def test_c5_full_size_update_vs_oracle_on_identical_post_augmentation_tensors():
    _run("c5 B=1024 168x168x12 L=6", (12, 168, 168), (168, 168), "color_jiggle", 6, 1024, False, capacity=1024)


def test_c1t_full_size_reference_shipped_geometry_update_vs_oracle():
    """Round 6: the ONE geometry the reference runs as shipped -- train.py's 90 x 160 camera frames (train.py:45-46)
    through the default RandomCrop (ceil(0.84 * side) = 76 x 135, augmentations.py:21-24) into the encoder's own shape
    table (encoder.py:26,42-43: 31 x 61 after four layers, fc over 60512 inputs) at train.py's batch size 512
    (train.py:72).  Rows of 67 / 65 / 63 / 61 pixels in the stride-1 stack; nothing about the shape is passed in: the
    augmentor computes it."""
    _run("c1t B=512 90x160->76x135 L=4", (9, 76, 135), (90, 160), "random_crop_default", 4, 512, False, capacity=1024)


def _steps_case(tag, obs_shape, in_hw, aug_name, capacity, B=512, layers=4):
    """The schedules of update() that the even step of _run() does not walk, at FULL batch, against the fp32 oracle
    differentiated along the device's ReLU branches (conv layers and twin-Q hidden units: values untouched, SURVEY.md
    D11), 1e-4 per tensor; all learning rates zero, so every phase of both sides sees the same weights:
      * an ODD step (curl_sac.py:436,441: no actor phase, no target update; update_cpc encodes anchor AND positives
        itself -- one two-problem launch per conv layer here, curl_sac.py:984-985 of this package);
      * ``update(..., only_cpc=True)`` (train.py:425-429): the CURL phase alone;
      * a GRAPH-REPLAYED odd and even step (enable_update_graphs): replayed from the captured hipGraph, then the same
        step eagerly from the same NumPy / Philox stream positions -- gradient buffers, features, logits, Q values
        and the policy noise bit for bit; the eager odd step is then held to the oracle with the noise the kernel
        drew (read back from the workspace), so the replayed update is pinned to the oracle through it."""
    from oracle import curla_oracle as O
    agent, oracle, rb, hp = _make(obs_shape, in_hw, aug_name, layers, B, False, capacity)
    dev = torch.device("cuda")
    out_hw = tuple(obs_shape[1:])
    kwc = dict(num_layers=layers, log_std_min=-10, log_std_max=2)
    ws = agent._ws(B)
    bad = []
    drawn = []
    real_sample = rb.sample_cpc_refs

    def sample():
        drawn.append(real_sample())
        return drawn[-1]
    rb.sample_cpc_refs = sample
    phases = []
    real_allreduce = agent._allreduce  # (called once per phase right in front of its optimizer step; no-op without DP)

    def tap(*buckets, **kw):
        if agent._graph_cap is None:  # (while a graph is being captured nothing may be copied to the host)
            phases.append(dict(
                critic=grads_of(agent.critic), W=agent.CURL.W.grad.detach().cpu().clone(),
                conv=[(a.permute(0, 3, 1, 2) > 0).cpu() for a in ws.acts_main],
                qh=[t.detach().cpu() > 0 for t in (ws.q_h1[0], ws.q_h2[0], ws.q_h1[1], ws.q_h2[1])]))
        return real_allreduce(*buckets, **kw)
    agent._allreduce = tap

    def oracle_soft_update():  # utils.py:37-41 with train.py's rates, as update() applies it on even steps
        with torch.no_grad():
            for prefix, tau in (("Q1.", hp["critic_tau"]), ("Q2.", hp["critic_tau"]), ("encoder.", hp["encoder_tau"])):
                O.soft_update(oracle.critic, oracle.critic_target, tau, prefix)

    def oracle_critic(p, sample_, noise, what):
        o_obs, o_nxt, _ = _host_minibatch(rb, sample_, aug_name, out_hw, capacity)
        _, act, rew, _, nd, _ = sample_
        r = O.critic_phase(oracle.actor, oracle.critic, oracle.critic_target, oracle.log_alpha, o_obs, act.cpu(),
                           rew.cpu(), o_nxt, nd.cpu(), noise.cpu(), discount=0.99, relu_branches=p["conv"],
                           q_branches=[p["qh"][0:2], p["qh"][2:4]], **kwc)
        assert len(r["grads"]) == 8 + 2 * layers + 8
        for k, v in r["grads"].items():
            bad.append(check(f"{tag} {what}: critic grad {k} (fp32 oracle along the device's branches)", p["critic"][k], v))
        return r

    def oracle_cpc(p, sample_, what, loss=None):
        o_obs, _, o_pos = _host_minibatch(rb, sample_, aug_name, out_hw, capacity)
        r = O.cpc_phase(oracle.critic, oracle.critic_target, oracle.W, o_obs, o_pos, num_layers=layers,
                        relu_branches=p["conv"])
        lg = ws.logits.cpu()
        lg = lg - lg.max(1, keepdim=True)[0]
        vals = [("cpc z_a", ws.z_c.cpu(), "z_a"), ("cpc z_pos", ws.z_pos.cpu(), "z_pos"),
                ("cpc logits (minus row max)", lg, "logits")]
        if loss is not None:
            vals.append(("curl loss", loss, "loss"))
        for nm, got, key in vals:
            bad.append(check(f"{tag} {what}: {nm}", got, r[key]))
        for k, v in r["grads"].items():
            bad.append(check(f"{tag} {what}: cpc grad {k} (fp32 oracle along the device's branches)", p["critic"][k], v))
        bad.append(check(f"{tag} {what}: cpc grad W", p["W"], r["W_grad"]))

    # ---- an odd step, eager, explicit noise
    before = agent._critic_flat.clone()
    L = NullLogger()
    noise_c = torch.randn(B, 2)
    agent.update(rb, L, 1, noise=(noise_c.to(dev), None))
    torch.cuda.synchronize()
    assert len(phases) == 2 and len(drawn) == 1  # critic, cpc
    rc = oracle_critic(phases[0], drawn[0], noise_c, "odd step")
    bad.append(check(f"{tag} odd step: critic loss", L.scalars["train_critic/loss"], rc["loss"]))
    oracle_cpc(phases[1], drawn[0], "odd step", L.scalars["train/curl_loss"])
    assert "train_actor/loss" not in L.scalars

    # ---- only_cpc (train.py:425-429)
    del phases[:], drawn[:]
    L = NullLogger()
    agent.update(rb, L, 2, only_cpc=True)
    torch.cuda.synchronize()
    assert len(phases) == 1 and list(L.scalars) == ["train/batch_reward", "train/curl_loss"], list(L.scalars)
    oracle_cpc(phases[0], drawn[0], "only_cpc", L.scalars["train/curl_loss"])
    assert torch.equal(before, agent._critic_flat), "learning rate 0: the parameters must not have moved"

    # ---- graph-replayed steps: log_interval off the steps used (logging steps run eagerly by design)
    agent.log_interval = 1000
    agent.enable_update_graphs(rb, warm=1, depth=1)
    gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
    agent.update(rb, L, 1001)   # warm: odd kind
    agent.update(rb, L, 1002)   # warm: even kind (its target soft update is applied to the oracle's target too)
    oracle_soft_update()

    def snapshot():
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in dict(
            gcritic=agent._critic_gflat, gactor=agent._actor_gflat, glog_alpha=agent.log_alpha.grad, z_c=ws.z_c,
            z_pos=ws.z_pos, z_a=ws.z_a, logits=ws.logits, q=ws.q2, noise=ws.noise, scalars=ws.scalars[:6],
            target=agent._target_flat, conv_last=ws.acts_main[-1]).items()}

    for step in (1003, 1004):
        what = "graph-replayed %s step" % ("odd" if step % 2 else "even")
        np_state, gen_off = np.random.get_state(), gen.get_offset()
        target0 = agent._target_flat.clone()
        del phases[:], drawn[:]
        agent.update(rb, L, step)
        replayed = snapshot()
        # nothing went through the host-side phases (the tap sits in the eager path): the update WAS a graph replay
        assert not phases and not drawn, (what, len(phases), len(drawn))
        kind = agent._graph_kind(step)
        assert kind in agent._graphs and agent._graphs[kind][0]["graph"] is not None
        # the same step again, eagerly, from the same stream positions (and the same target: an even step lerps it)
        np.random.set_state(np_state)
        gen.set_offset(gen_off)
        with torch.no_grad():
            agent._target_flat.copy_(target0)
        graphs, agent._graphs = agent._graphs, None
        agent.update(rb, L, step)
        eager = snapshot()
        agent._graphs = graphs
        for k in replayed:
            assert torch.equal(replayed[k], eager[k]), (what, k)
        REPORT.append((f"{tag} {what}: buffers bit-identical to the eager step (count)", float(len(replayed))))
        if step % 2:  # the eager twin against the oracle, with the policy noise the kernel drew
            assert len(phases) == 2
            oracle_critic(phases[0], drawn[0], ws.noise.cpu().clone(), what)
            oracle_cpc(phases[1], drawn[0], what)
        else:
            assert len(phases) == 3
            oracle_soft_update()
            for k in ("encoder.convs.0.weight", "encoder.fc.weight", "Q1.trunk.2.weight"):
                bad.append(check(f"{tag} {what}: target after soft update {k}", agent.critic_target.state_dict()[k].cpu(),
                                 oracle.critic_target[k].detach(), 1e-6))
            oracle_cpc(phases[2], drawn[0], what)
    assert torch.equal(before, agent._critic_flat)
    bad = [b for b in bad if b is not None]
    assert not bad, bad


def test_c2_full_size_odd_only_cpc_and_graph_replayed_steps_vs_oracle():
    _steps_case("c2 B=512 84->76 L=4", (9, 76, 76), (84, 84), "random_crop", capacity=2048)


def test_c1t_full_size_odd_only_cpc_and_graph_replayed_steps_vs_oracle():
    _steps_case("c1t B=512 90x160->76x135 L=4", (9, 76, 135), (90, 160), "random_crop_default", capacity=1024)
