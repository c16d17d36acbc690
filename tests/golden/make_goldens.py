#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED
reference (paulvantieghem/curla, mounted read-only at /root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box);
only the .npz vectors it writes are committed.  Recipe:

* three third-party imports the reference needs but the image lacks are
  replaced by ``sys.modules`` stand-ins that carry no arithmetic of the path:
  ``gymnasium`` (Wrapper/spaces only used by FrameStack, off-path),
  ``kornia.augmentation`` (constructors only; colour jiggle is not exercised),
  ``skimage.util.shape.view_as_windows`` -> ``numpy.lib.stride_tricks.
  sliding_window_view`` (same strided view; the crop result is a byte gather);
* ``encoder.OUT_DIM`` (a module global) is set so ``CNNEncoder`` accepts the
  crop sizes used here (the shipped table only has 84/64 squares and two
  rectangles, encoder.py:21-47);
* ``RandomCrop.output_shape`` is overridden per fixture (the shipped factor
  0.84 maps 84->71, augmentations.py:23-24; BASELINE.json asks for 84->76).

Fixtures:
  tiny.npz      9x34x40 -> crop 28x34 (rectangular on purpose), hidden 64, B=8: full tensors of one
                even-step update() after 4 warm-up updates from the reference's
                own init (state, batch, crops, noise, per-phase activations,
                losses, gradients, post-step snapshots) + acting path.
  tiny_rank1.npz .. tiny_rank7.npz  the next seven minibatches, each from the same
                 pre-update state (for the 2-, 4- and 8-rank data-parallel
                 mean-of-gradients definition).
  crop84.npz    random_crop / center-crop bytes + index stream at 84->76.
  c1shape.npz   9x84x84 -> 76x76, hidden 128, B=4, weights from a NumPy
                recipe (regenerable from seed): losses + per-tensor gradient
                summaries of one update.
  mode_*.npz    (round 5) the branches of update() that tiny.npz does not walk,
                one whole reference update() each from seeded weights and fresh
                optimizers (tests/golden_recipes.py: MODES):
                  mode_odd       step 1: no actor phase, no target update (curl_sac.py:436,441)
                  mode_pixel_sac pixel_sac=True: no CURL phase (curl_sac.py:448)
                  mode_only_cpc  update(..., only_cpc=True) (train.py:425-429)
                  mode_detach    detach_encoder=True (curl_sac.py:358)
                  mode_l6c12     num_layers=6, 12 input channels, identity augmentation
                                 (encoder.py:54-63; utils.py:168-182): configs[4]'s geometry, small
                  mode_thesis, mode_thesis_odd  (round 6) the reference AS SHIPPED: 90x160 frames (train.py:45-46),
                                 the default RandomCrop factor 0.84 -> 76x135 (augmentations.py:21-24) and the
                                 encoder's own shape table (encoder.py:26,42-43) -- encoder.OUT_DIM is NOT assigned
                                 and RandomCrop.output_shape is NOT overridden for these two; B=4, an even and an
                                 odd update()
                every gradient an optimizer consumed (full tensors), the logged
                scalars, the RNG draws / noise, and the parameters after the
                update (small tensors whole, the first 10000 elements of big ones).
"""
import copy
import hashlib
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("CURLA_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    gym = types.ModuleType("gymnasium")

    class Wrapper:  # only FrameStack (off-path) derives from it
        def __init__(self, env):
            self.env = env
    gym.Wrapper = Wrapper
    gym.spaces = types.SimpleNamespace(Box=lambda **kw: None)
    sys.modules["gymnasium"] = gym

    sk = types.ModuleType("skimage")
    sku = types.ModuleType("skimage.util")
    skus = types.ModuleType("skimage.util.shape")
    skus.view_as_windows = lambda arr, shape: np.lib.stride_tricks.sliding_window_view(arr, shape)
    sk.util, sku.shape = sku, skus
    sys.modules.update({"skimage": sk, "skimage.util": sku, "skimage.util.shape": skus})

    ko = types.ModuleType("kornia")
    ka = types.ModuleType("kornia.augmentation")
    ka.ColorJiggle = lambda **kw: None
    ka.RandomGaussianNoise = lambda **kw: None
    ko.augmentation = ka
    sys.modules.update({"kornia": ko, "kornia.augmentation": ka})


_install_stubs()
sys.path.insert(0, REF)
import torch  # noqa: E402

torch.set_num_threads(1)  # bit-reproducible fixtures
import augmentations  # noqa: E402
import curl_sac  # noqa: E402
import encoder  # noqa: E402
import utils  # noqa: E402

sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.curla_oracle import conv_out_hw  # noqa: E402  (shape arithmetic only)
from tests.golden_recipes import fill_transitions, numpy_weights  # noqa: E402  (seeded inputs)


class NullLogger:
    def __init__(self):
        self.scalars = {}

    def log(self, key, value, step, n=1):
        if isinstance(value, torch.Tensor):
            value = value.item()
        self.scalars[key] = float(value)

    def log_histogram(self, *a, **k):
        pass

    def log_param(self, *a, **k):
        pass

    def log_image(self, *a, **k):
        pass


def make_crop_augmentor(in_hw, out_hw):
    aug = augmentations.RandomCrop(in_hw)
    aug.output_shape = tuple(out_hw)
    return aug


def fill_buffer(rb, n, obs_shape, seed):
    for obs, act, rew, nxt, done in fill_transitions(n, obs_shape, seed):
        rb.add(obs, act, rew, nxt, done)


def sd_np(prefix, sd):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


class Recorder:
    """Wraps RNG draws and module calls of ONE reference update()."""

    def __init__(self, agent):
        self.agent = agent
        self.rec = {}
        self.randint_calls = []
        self.noises = []

    def run(self, rb, step, only_cpc=False):
        agent, rec = self.agent, self.rec
        L = NullLogger()
        orig_randint, orig_randn_like = np.random.randint, torch.randn_like

        def randint(*a, **k):
            r = orig_randint(*a, **k)
            self.randint_calls.append(np.array(r).copy())
            return r

        def randn_like(t, **k):
            r = orig_randn_like(t, **k)
            self.noises.append(r.detach().numpy().copy())
            return r

        def wrap_step(opt, name, params_named, after=None):
            orig = opt.step

            def step_(*a, **k):
                for n_, p_ in params_named:
                    if p_.grad is not None:
                        rec[f"{name}/grad/{n_}"] = p_.grad.detach().numpy().copy()
                out = orig(*a, **k)
                if after is not None:
                    after()
                return out
            opt.step = step_
            return orig

        critic_named = list(agent.critic.named_parameters())
        actor_named = [(n_, p_) for n_, p_ in agent.actor.named_parameters()]
        enc_named = [("encoder." + n_, p_) for n_, p_ in agent.critic.encoder.named_parameters()]

        def after_critic():
            rec.update(sd_np("critic_after/", agent.critic.state_dict()))

        def after_actor():
            rec["actor_after/trunk.4.weight"] = agent.actor.state_dict()["trunk.4.weight"].numpy().copy()

        origs = [
            (agent.critic_optimizer, wrap_step(agent.critic_optimizer, "critic", critic_named, after_critic)),
            (agent.actor_optimizer, wrap_step(agent.actor_optimizer, "actor", actor_named, after_actor)),
            (agent.log_alpha_optimizer, wrap_step(agent.log_alpha_optimizer, "alpha", [("log_alpha", agent.log_alpha)])),
            (agent.encoder_optimizer, wrap_step(agent.encoder_optimizer, "cpc", enc_named + [("W", agent.CURL.W)])),
        ]

        # module forwards, recorded in call order
        calls = {"actor": [], "critic": [], "critic_target": []}

        def wrap_forward(mod, name):
            orig = mod.forward

            def fwd(*a, **k):
                out = orig(*a, **k)
                calls[name].append(tuple(None if o is None else o.detach().numpy().copy() for o in out))
                if name == "critic" and len(calls[name]) == 1:
                    for kk, vv in mod.encoder.outputs.items():
                        if kk != "obs":
                            rec[f"critic/enc/{kk}"] = vv.detach().numpy().copy()
                return out
            mod.forward = fwd
            return orig
        f_orig = [(m, wrap_forward(m, n_)) for m, n_ in ((agent.actor, "actor"), (agent.critic, "critic"),
                                                          (agent.critic_target, "critic_target"))]
        orig_logits = agent.CURL.compute_logits

        def compute_logits(z_a, z_pos):
            rec["cpc/z_a"] = z_a.detach().numpy().copy()
            rec["cpc/z_pos"] = z_pos.detach().numpy().copy()
            out = orig_logits(z_a, z_pos)
            rec["cpc/logits"] = out.detach().numpy().copy()
            return out
        agent.CURL.compute_logits = compute_logits
        orig_cpc = agent.update_cpc

        def update_cpc(obs_anchor, obs_pos, cpc_kwargs, L_, step_):
            rec.update(sd_np("target_after/", agent.critic_target.state_dict()))
            rec["batch/pos"] = obs_pos.numpy().astype(np.uint8)
            if "batch/obs" not in rec:  # (only_cpc: update_critic, which records the batch, does not run)
                rec["batch/obs"] = obs_anchor.numpy().astype(np.uint8)
            return orig_cpc(obs_anchor, obs_pos, cpc_kwargs, L_, step_)
        agent.update_cpc = update_cpc
        orig_uc = agent.update_critic

        def update_critic(obs, action, reward, next_obs, not_done, L_, step_):
            rec["batch/obs"] = obs.numpy().astype(np.uint8)
            rec["batch/next_obs"] = next_obs.numpy().astype(np.uint8)
            rec["batch/action"] = action.numpy().copy()
            rec["batch/reward"] = reward.numpy().copy()
            rec["batch/not_done"] = not_done.numpy().copy()
            return orig_uc(obs, action, reward, next_obs, not_done, L_, step_)
        agent.update_critic = update_critic

        np.random.randint, torch.randn_like = randint, randn_like
        try:
            if only_cpc:
                agent.update(rb, L, step, only_cpc=True)
            else:
                agent.update(rb, L, step)
        finally:
            np.random.randint, torch.randn_like = orig_randint, orig_randn_like
            for opt, o in origs:
                opt.step = o
            for m, o in f_orig:
                m.forward = o
            agent.CURL.compute_logits = orig_logits
            agent.update_cpc, agent.update_critic = orig_cpc, orig_uc

        # RNG draws: idxs, then (h1, w1) x3
        rc = self.randint_calls
        rec["rng/idxs"] = rc[0]
        if len(rc) >= 7:  # (an identity augmentation draws the indices only)
            for j, nm in enumerate(("obs", "next_obs", "pos")):
                rec[f"rng/h1_{nm}"], rec[f"rng/w1_{nm}"] = rc[1 + 2 * j], rc[2 + 2 * j]
        rec["batch/obs_full"] = rb.obses[rc[0]].copy()
        rec["batch/next_obs_full"] = rb.next_obses[rc[0]].copy()
        if len(self.noises) >= 1:
            rec["noise/critic"] = self.noises[0]
        if len(self.noises) >= 2:
            rec["noise/actor"] = self.noises[1]
        # forwards: actor#0 = actor(next_obs) [critic phase], actor#1 = actor(obs) [actor phase]
        if len(calls["actor"]) >= 1:
            a0 = calls["actor"][0]
            rec["critic/policy_action"], rec["critic/next_log_pi"] = a0[1], a0[2]
            rec["critic/tq1"], rec["critic/tq2"] = calls["critic_target"][0]
            rec["critic/q1"], rec["critic/q2"] = calls["critic"][0]
        if len(calls["actor"]) >= 2:
            a1 = calls["actor"][1]
            rec["actor/mu"], rec["actor/pi"], rec["actor/log_pi"], rec["actor/log_std"] = a1
            rec["actor/q1"], rec["actor/q2"] = calls["critic"][1]
        for k, v in L.scalars.items():
            rec["scalar/" + k] = np.float64(v)
        return rec


SHIPPED_OUT_DIM = dict(encoder.OUT_DIM)  # encoder.py:21, before any fixture touches it


def build_agent(obs_shape_key, real_hw, hidden, num_layers, seed, aug, detach_encoder=False, pixel_sac=False,
                unpatched=False):
    if unpatched:  # the shipped shape tables decide (encoder.py:38-47); nothing of the reference is assigned to
        encoder.OUT_DIM = dict(SHIPPED_OUT_DIM)
        assert tuple(obs_shape_key[1:]) == tuple(real_hw)
    else:
        encoder.OUT_DIM = {num_layers: list(conv_out_hw(real_hw[0], real_hw[1], num_layers))}
    utils.set_seed_everywhere(seed)
    agent = curl_sac.CurlSacAgent(
        obs_shape=obs_shape_key, action_shape=(2,), device=torch.device("cpu"), augmentor=aug,
        hidden_dim=hidden, discount=0.99, init_temperature=0.1, alpha_lr=1e-4, alpha_beta=0.5,
        actor_lr=1e-3, actor_beta=0.9, actor_log_std_min=-10, actor_log_std_max=2, actor_update_freq=2,
        critic_lr=1e-3, critic_beta=0.9, critic_tau=0.01, critic_target_update_freq=2,
        encoder_feature_dim=50, encoder_lr=1e-3, encoder_tau=0.05, num_layers=num_layers, num_filters=32,
        log_interval=1, log_param_hist_imgs=False, detach_encoder=detach_encoder, pixel_sac=pixel_sac)
    if not unpatched:
        agent.image_shape = tuple(real_hw)
    return agent


def state_np(agent):
    st = {}
    st.update(sd_np("state0/actor/", agent.actor.state_dict()))
    st.update(sd_np("state0/critic/", agent.critic.state_dict()))
    st.update(sd_np("state0/critic_target/", agent.critic_target.state_dict()))
    st["state0/W"] = agent.CURL.W.detach().numpy().copy()
    st["state0/log_alpha"] = agent.log_alpha.detach().numpy().copy()
    return st


def gen_tiny():
    in_hw, out_hw, B = (34, 40), (28, 34), 8
    aug = make_crop_augmentor(in_hw, out_hw)
    agent = build_agent((9, 84, 84), out_hw, hidden=64, num_layers=4, seed=1, aug=aug)
    # construction parity (CurlSacAgent.__init__ + weight_init, curl_sac.py:38-54,226-318): summaries of the
    # freshly initialised parameters for seed 1
    init = {}
    for name, mod in (("actor", agent.actor), ("critic", agent.critic), ("critic_target", agent.critic_target)):
        for k, v in mod.state_dict().items():
            init[f"{name}/{k}"] = summarize(v.numpy())
    init["W"] = summarize(agent.CURL.W.detach().numpy())
    init["log_alpha"] = agent.log_alpha.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "init_tiny.npz"), **init)
    rb = utils.ReplayBuffer((9,) + in_hw, (2,), 64, B, torch.device("cpu"), aug)
    fill_buffer(rb, 40, (9,) + in_hw, seed=0)
    L = NullLogger()
    for step in range(4):  # warm-up: move the convs off the delta-orthogonal init
        agent.update(rb, L, step)
    for m in (agent.actor, agent.critic, agent.critic_target):
        m.outputs.clear()          # cached non-leaf activations block deepcopy
        m.encoder.outputs.clear()
    agent0 = copy.deepcopy(agent)
    np_state, torch_state = np.random.get_state(), torch.get_rng_state()

    rec = state_np(agent)
    rec.update(Recorder(agent).run(rb, 4))
    rec["meta/in_hw"], rec["meta/out_hw"] = np.array(in_hw), np.array(out_hw)
    rec["meta/n_valid"] = np.int64(40)
    rec["meta/numpy_state_keys"] = np_state[1]          # MT19937 key vector before the update
    rec["meta/numpy_state_pos"] = np.int64(np_state[2])
    # post-update scalars (after the cpc Adam steps)
    rec["final/W"] = agent.CURL.W.detach().numpy().copy()
    rec["final/log_alpha"] = agent.log_alpha.detach().numpy().copy()

    # acting path from state0 (curl_sac.py:330-347)
    rs = np.random.RandomState(7)
    eval_obs = rs.randint(0, 256, (9,) + in_hw, dtype=np.uint8)
    rec["act/obs"] = eval_obs
    rec["act/select"] = agent0.select_action(aug.evaluation_augmentation(eval_obs))
    noises = []
    orig = torch.randn_like

    def randn_like(t, **k):
        r = orig(t, **k)
        noises.append(r.numpy().copy())
        return r
    torch.randn_like = randn_like
    try:
        rec["act/sample"] = agent0.sample_action(eval_obs)
    finally:
        torch.randn_like = orig
    rec["act/noise"] = noises[0]
    np.savez_compressed(os.path.join(HERE, "tiny.npz"), **rec)

    # ranks 1..7 of the N-rank DP definition (N = 2 and 8): every rank starts from the same state0
    # (an update() mutates its agent, so each rank gets its own copy, taken before any of them runs) and draws the
    # NEXT minibatch/noise in the streams
    replicas = [agent0] + [copy.deepcopy(agent0) for _ in range(6)]
    for r, replica in enumerate(replicas, start=1):
        rec_r = Recorder(replica).run(rb, 4)
        keep = {k: v for k, v in rec_r.items() if k.split("/")[0] in ("batch", "rng", "noise", "scalar")
                or "/grad/" in k}
        np.savez_compressed(os.path.join(HERE, "tiny_rank%d.npz" % r), **keep)
    return rec


def gen_crop84():
    aug = make_crop_augmentor((84, 84), (76, 76))
    rs = np.random.RandomState(3)
    imgs = rs.randint(0, 256, (16, 9, 84, 84), dtype=np.uint8)
    np.random.seed(1234)
    calls = []
    orig = np.random.randint

    def randint(*a, **k):
        r = orig(*a, **k)
        calls.append(np.array(r).copy())
        return r
    np.random.randint = randint
    try:
        out = aug.training_augmentation(imgs)
    finally:
        np.random.randint = orig
    out = np.ascontiguousarray(out)
    cc = aug.evaluation_augmentation(imgs[0])
    # default-factor augmentor (0.84 -> 71) for the ctor arithmetic
    dflt = augmentations.RandomCrop((84, 84)).output_shape
    dflt_rect = augmentations.RandomCrop((90, 160)).output_shape
    np.savez_compressed(os.path.join(HERE, "crop84.npz"), imgs_seed=np.int64(3), numpy_seed=np.int64(1234),
                        h1=calls[0], w1=calls[1], out_sha256=np.array(hashlib.sha256(out.tobytes()).hexdigest()),
                        out_first2=out[:2], center_crop0=np.ascontiguousarray(cc),
                        default_shape_84=np.array(dflt), default_shape_90_160=np.array(dflt_rect))


def summarize(t):
    t = np.asarray(t, dtype=np.float64).ravel()
    n = t.size
    idx = (np.arange(16) * max(1, n // 16)) % n
    return np.concatenate([[n, t.sum(), np.abs(t).sum(), np.sqrt((t * t).sum())], t[:16 if n >= 16 else n],
                           t[idx]])


def gen_c1shape():
    in_hw, out_hw, B = (84, 84), (76, 76), 4
    aug = make_crop_augmentor(in_hw, out_hw)
    agent = build_agent((9, 84, 84), out_hw, hidden=128, num_layers=4, seed=1, aug=aug)
    with torch.no_grad():
        for mod, seed in ((agent.critic, 11), (agent.actor, 12)):
            shapes = [(k, tuple(v.shape)) for k, v in mod.named_parameters()]
            w = numpy_weights(shapes, seed)
            for k, v in mod.named_parameters():
                v.copy_(torch.from_numpy(w[k]))
        agent.critic_target.load_state_dict(agent.critic.state_dict())
        # make the target differ from the online net, as in training
        shapes = [(k, tuple(v.shape)) for k, v in agent.critic_target.named_parameters()]
        w = numpy_weights(shapes, 13)
        for k, v in agent.critic_target.named_parameters():
            v.mul_(0.9).add_(0.1 * torch.from_numpy(w[k]))
        agent.CURL.W.copy_(torch.from_numpy(np.random.RandomState(14).rand(50, 50).astype(np.float32)))
    rb = utils.ReplayBuffer((9,) + in_hw, (2,), 16, B, torch.device("cpu"), aug)
    fill_buffer(rb, 16, (9,) + in_hw, seed=5)
    np.random.seed(99)
    torch.manual_seed(99)
    rec = Recorder(agent).run(rb, 0)
    out = {}
    for k, v in rec.items():
        top = k.split("/")[0]
        if top in ("rng", "noise", "scalar") or k in ("batch/action", "batch/reward", "batch/not_done"):
            out[k] = v
        elif "/grad/" in k or k in ("critic/q1", "critic/q2", "critic/tq1", "critic/tq2", "actor/pi",
                                     "actor/log_pi", "cpc/z_a", "cpc/z_pos", "cpc/logits"):
            out["sum/" + k] = summarize(v)
    out["batch/obs_sha256"] = np.array(hashlib.sha256(rec["batch/obs"].tobytes()).hexdigest())
    out["meta/buffer_seed"], out["meta/numpy_seed"] = np.int64(5), np.int64(99)
    out["meta/weight_seeds"] = np.array([11, 12, 13, 14])
    np.savez_compressed(os.path.join(HERE, "c1shape.npz"), **out)


def gen_noisy_cover():
    """NoisyCover.training_augmentation: the cover rows / colours / clamp are the reference's own code;
    kornia's RandomGaussianNoise (absent) is replaced by a recorded additive-noise stand-in, so this
    pins everything except the noise distribution."""
    aug = augmentations.NoisyCover((34, 40))
    rs = np.random.RandomState(21)
    imgs = rs.randint(0, 256, (5, 9, 34, 40), dtype=np.uint8)
    noise = torch.from_numpy(rs.randn(5, 9, 34, 40).astype(np.float32) * 10.0)
    aug.aug = lambda x: x + noise.reshape(x.shape)
    np.random.seed(77)
    out = aug.training_augmentation(torch.from_numpy(imgs).float())
    np.random.seed(77)
    colors = np.array([np.random.randint(0, 255) for _ in range(3)])
    np.savez_compressed(os.path.join(HERE, "noisy_cover.npz"), imgs_seed=np.int64(21), numpy_seed=np.int64(77),
                        colors=colors, top=np.int64(aug.top), bottom=np.int64(aug.bottom),
                        out=out.numpy().astype(np.float16), out_sum=np.float64(out.double().sum().item()))


def apply_recipe_weights(agent, seeds):
    """Weights from tests/golden_recipes.numpy_weights, as gen_c1shape applies them (= golden_recipes.synthetic_state)."""
    with torch.no_grad():
        for mod, seed in ((agent.critic, seeds[0]), (agent.actor, seeds[1])):
            shapes = [(k, tuple(v.shape)) for k, v in mod.named_parameters()]
            w = numpy_weights(shapes, seed)
            for k, v in mod.named_parameters():
                v.copy_(torch.from_numpy(w[k]))
        agent.critic_target.load_state_dict(agent.critic.state_dict())
        shapes = [(k, tuple(v.shape)) for k, v in agent.critic_target.named_parameters()]
        w = numpy_weights(shapes, seeds[2])
        for k, v in agent.critic_target.named_parameters():
            v.mul_(0.9).add_(0.1 * torch.from_numpy(w[k]))
        agent.CURL.W.copy_(torch.from_numpy(np.random.RandomState(seeds[3]).rand(50, 50).astype(np.float32)))


POST_CLIP = 10000  # elements kept of a parameter after the update (tests/golden_recipes.py uses the same number)


def gen_modes():
    """One whole reference update() per mode from seeded weights and FRESH optimizers (so the oracle's and the HIP
    agent's own Adam steps can follow it to the post-update parameters)."""
    from tests.golden_recipes import BIG, MODES, big_sample
    for name, m in MODES.items():
        c, in_hw, out_hw, layers = m["channels"], tuple(m["in_hw"]), tuple(m["out_hw"]), m["num_layers"]
        unpatched = bool(m.get("unpatched"))
        if unpatched:
            # the reference exactly as shipped: default RandomCrop (augmentations.py:21-24), shipped encoder tables
            aug = augmentations.RandomCrop(in_hw)
            assert tuple(aug.output_shape) == out_hw, (aug.output_shape, out_hw)
        elif m["crop"]:
            aug = make_crop_augmentor(in_hw, out_hw)
        else:
            aug = augmentations.IdentityAugmentation(in_hw)
        agent = build_agent((c,) + out_hw if unpatched else (c, 84, 84), out_hw, hidden=m["hidden"], num_layers=layers,
                            seed=1, aug=aug, detach_encoder=m["detach_encoder"], pixel_sac=m["pixel_sac"],
                            unpatched=unpatched)
        apply_recipe_weights(agent, m["weight_seeds"])
        rb = utils.ReplayBuffer((c,) + in_hw, (2,), m["n_fill"], m["batch"], torch.device("cpu"), aug)
        fill_buffer(rb, m["n_fill"], (c,) + in_hw, seed=m["buffer_seed"])
        np.random.seed(m["numpy_seed"])
        torch.manual_seed(m["numpy_seed"])
        recorder = Recorder(agent)
        rec = recorder.run(rb, m["step"], only_cpc=m["only_cpc"])
        out = {"rng/n_draws": np.int64(len(recorder.randint_calls))}
        for k, v in rec.items():
            top = k.split("/")[0]
            if "/grad/" in k and np.size(v) > BIG:
                # (fc.weight at 76 x 135 is 50 x 60512: a strided sample of its elements + a summary of the whole)
                out[k] = big_sample(v)
                out[k.replace("/grad/", "/gradsum/")] = summarize(v)
            elif top in ("rng", "noise", "scalar") or "/grad/" in k or k in (
                    "batch/action", "batch/reward", "batch/not_done", "critic/q1", "critic/q2", "critic/tq1",
                    "critic/tq2", "actor/pi", "actor/log_pi", "cpc/logits"):
                out[k] = v
        for key in ("batch/obs", "batch/next_obs", "batch/pos"):
            if key in rec:
                out[key + "_sha256"] = np.array(hashlib.sha256(rec[key].tobytes()).hexdigest())
        for tag, sd in (("actor", agent.actor.state_dict()), ("critic", agent.critic.state_dict()),
                        ("critic_target", agent.critic_target.state_dict())):
            for k, v in sd.items():
                out[f"post/{tag}/{k}"] = v.detach().numpy().ravel()[:POST_CLIP].copy()
        out["post/W"] = agent.CURL.W.detach().numpy().copy()
        out["post/log_alpha"] = agent.log_alpha.detach().numpy().copy()
        for tag, opt in (("critic", agent.critic_optimizer), ("actor", agent.actor_optimizer),
                         ("alpha", agent.log_alpha_optimizer), ("encoder", agent.encoder_optimizer),
                         ("cpc", agent.cpc_optimizer)):
            steps = [int(st["step"]) for st in opt.state.values() if "step" in st]
            out[f"post/adam_steps/{tag}"] = np.array([len(steps), max(steps) if steps else 0], dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, "mode_%s.npz" % name), **out)


if __name__ == "__main__":
    if "--modes-only" in sys.argv:
        gen_modes()
        sys.exit(0)
    gen_noisy_cover()
    gen_tiny()
    gen_crop84()
    gen_c1shape()
    gen_modes()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
