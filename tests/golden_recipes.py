"""Seed-regenerable inputs shared by tests/golden/make_goldens.py (which applies
them to the reference) and the parity tests (which apply them to the oracle and
to the HIP path).  Nothing here touches /root/reference."""
import numpy as np
import torch

from oracle import curla_oracle as O


def fill_transitions(n, obs_shape, seed):
    """Synthetic replay contents: i.i.d. uniform bytes,
    action ~ U(-1,1)^2, reward ~ N(0,1), done every 50th transition."""
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        obs = rs.randint(0, 256, obs_shape, dtype=np.uint8)
        nxt = rs.randint(0, 256, obs_shape, dtype=np.uint8)
        act = rs.uniform(-1, 1, 2).astype(np.float32)
        rew = np.float32(rs.randn())
        out.append((obs, act, rew, nxt, (i % 50) == 49))
    return out


def numpy_weights(shapes, seed):
    """Weights regenerable from a seed, drawn in the order given."""
    rs = np.random.RandomState(seed)
    out = {}
    for k, shp in shapes:
        if k.endswith("ln.weight"):
            out[k] = (1.0 + 0.1 * rs.randn(*shp)).astype(np.float32)
        elif k.endswith("bias"):
            out[k] = (0.05 * rs.randn(*shp)).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            out[k] = (rs.randn(*shp) * (1.5 / np.sqrt(fan_in))).astype(np.float32)
    return out


def encoder_shapes(prefix, in_ch, hw, feat, num_layers, nf):
    s = []
    for i in range(num_layers):
        s += [(f"{prefix}convs.{i}.weight", (nf, in_ch if i == 0 else nf, 3, 3)), (f"{prefix}convs.{i}.bias", (nf,))]
    oh, ow = O.conv_out_hw(hw[0], hw[1], num_layers)
    s += [(f"{prefix}fc.weight", (feat, nf * oh * ow)), (f"{prefix}fc.bias", (feat,)),
          (f"{prefix}ln.weight", (feat,)), (f"{prefix}ln.bias", (feat,))]
    return s


def mlp_shapes(prefix, din, hidden, dout):
    return [(f"{prefix}0.weight", (hidden, din)), (f"{prefix}0.bias", (hidden,)),
            (f"{prefix}2.weight", (hidden, hidden)), (f"{prefix}2.bias", (hidden,)),
            (f"{prefix}4.weight", (dout, hidden)), (f"{prefix}4.bias", (dout,))]


def critic_shapes(in_ch, hw, feat, num_layers, nf, hidden, act_dim):
    """named_parameters() order of the reference Critic (curl_sac.py:142-156)."""
    return (encoder_shapes("encoder.", in_ch, hw, feat, num_layers, nf)
            + mlp_shapes("Q1.trunk.", feat + act_dim, hidden, 1) + mlp_shapes("Q2.trunk.", feat + act_dim, hidden, 1))


def actor_shapes(in_ch, hw, feat, num_layers, nf, hidden, act_dim):
    """named_parameters() order of the reference Actor (curl_sac.py:57-77)."""
    return encoder_shapes("encoder.", in_ch, hw, feat, num_layers, nf) + mlp_shapes("trunk.", feat, hidden, 2 * act_dim)


def synthetic_state(in_ch, hw, feat=50, num_layers=4, nf=32, hidden=128, act_dim=2, seeds=(11, 12, 13, 14)):
    """(actor, critic, critic_target, W) as make_goldens.gen_c1shape builds them:
    critic from seeds[0]; actor from seeds[1] (its convs are the tied tensors, so
    they overwrite the critic's); target = 0.9*critic + 0.1*seeds[2]; W from seeds[3]."""
    cs = critic_shapes(in_ch, hw, feat, num_layers, nf, hidden, act_dim)
    as_ = actor_shapes(in_ch, hw, feat, num_layers, nf, hidden, act_dim)
    critic = numpy_weights(cs, seeds[0])
    actor_full = numpy_weights(as_, seeds[1])
    for k in actor_full:
        if ".convs." in k:
            critic[k] = actor_full[k]
    actor = {k: v for k, v in actor_full.items() if ".convs." not in k}
    noise = numpy_weights(cs, seeds[2])
    target = {k: (critic[k] * np.float32(0.9) + np.float32(0.1) * noise[k]).astype(np.float32) for k in critic}
    W = np.random.RandomState(seeds[3]).rand(feat, feat).astype(np.float32)
    t = lambda d: {k: torch.from_numpy(v.copy()) for k, v in d.items()}  # noqa: E731
    return t(actor), t(critic), t(target), torch.from_numpy(W)


def c1shape_inputs(g):
    """Rebuild the c1shape fixture's inputs (84x84x9 -> 76x76, B=4, hidden 128)."""
    trans = fill_transitions(16, (9, 84, 84), int(g["meta/buffer_seed"]))
    obses = np.stack([t[0] for t in trans])
    nexts = np.stack([t[3] for t in trans])
    idxs = g["rng/idxs"]
    crop = lambda src, nm: O.random_crop(src[idxs], g[f"rng/h1_{nm}"], g[f"rng/w1_{nm}"], (76, 76))  # noqa: E731
    actor, critic, target, W = synthetic_state(9, (76, 76), seeds=tuple(int(s) for s in g["meta/weight_seeds"]))
    agent = O.OracleAgent((9, 76, 76), (2,), hidden_dim=128)
    for dst, src in ((agent.actor, actor), (agent.critic, critic), (agent.critic_target, target)):
        for k in dst:
            dst[k].data.copy_(src[k])
    agent.W.data.copy_(W)
    return dict(obs=crop(obses, "obs"), next_obs=crop(nexts, "next_obs"), pos=crop(obses, "pos"),
                obs_full=obses[idxs], next_obs_full=nexts[idxs],
                actor=actor, critic=critic, target=target, W=W, log_alpha=torch.tensor(np.log(0.1)), agent=agent)


# ----------------------------------------------------------------------------------------------------------------
# round 5: the branches of update() beside the even CURL step (tests/golden/mode_*.npz; make_goldens.gen_modes applies
# these recipes to the reference, the parity tests to the oracle and to the HIP agent)
_MODE = dict(channels=9, in_hw=(26, 30), out_hw=(22, 26), crop=True, num_layers=4, hidden=32, batch=8, n_fill=16,
             step=0, only_cpc=False, detach_encoder=False, pixel_sac=False, buffer_seed=5, numpy_seed=99,
             weight_seeds=(11, 12, 13, 14))
MODES = {
    "odd": dict(_MODE, step=1, numpy_seed=101),
    "pixel_sac": dict(_MODE, pixel_sac=True, numpy_seed=102),
    "only_cpc": dict(_MODE, only_cpc=True, numpy_seed=103),
    "detach": dict(_MODE, detach_encoder=True, numpy_seed=104),
    "l6c12": dict(_MODE, channels=12, in_hw=(30, 32), out_hw=(30, 32), crop=False, num_layers=6, numpy_seed=105),
    # round 6: the ONE geometry the shipped reference runs with no harness edit at all -- train.py's 90 x 160 camera
    # frames (train.py:45-46) through the DEFAULT RandomCrop (factor 0.84 -> 76 x 135, augmentations.py:21-24) into the
    # encoder's SHIPPED shape table (encoder.py:26,42-43: [31, 61]).  make_goldens.gen_modes neither assigns
    # encoder.OUT_DIM nor overrides RandomCrop.output_shape for these two ("unpatched"); an even and an odd update().
    "thesis": dict(_MODE, in_hw=(90, 160), out_hw=(76, 135), batch=4, n_fill=8, numpy_seed=106, unpatched=True),
    "thesis_odd": dict(_MODE, in_hw=(90, 160), out_hw=(76, 135), batch=4, n_fill=8, step=1, numpy_seed=107,
                       unpatched=True),
}
POST_CLIP = 10000
BIG = 1 << 18   # a gradient tensor with more elements is stored as a strided sample (+ a summary of the whole)


def big_sample(v):
    """The strided sample make_goldens keeps of a gradient with more than BIG elements (~32k of them, reference
    element order): the whole tensor for anything smaller."""
    v = np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v)
    if v.size <= BIG:
        return v
    return np.ascontiguousarray(v.ravel()[::-(-v.size // 32768)])


def mode_inputs(name, g):
    """Rebuild a mode fixture's inputs from its seeds and recorded draws: frames of the whole buffer, the minibatch's
    crops, the seeded parameters."""
    m = MODES[name]
    c, in_hw, out_hw = m["channels"], tuple(m["in_hw"]), tuple(m["out_hw"])
    trans = fill_transitions(m["n_fill"], (c,) + in_hw, m["buffer_seed"])
    obses, nexts = np.stack([t[0] for t in trans]), np.stack([t[3] for t in trans])
    acts = np.stack([t[1] for t in trans])
    rews = np.array([t[2] for t in trans], dtype=np.float32)
    dones = np.array([t[4] for t in trans])
    idxs = g["rng/idxs"]
    if m["crop"]:
        crop = lambda src, nm: O.random_crop(src[idxs], g[f"rng/h1_{nm}"], g[f"rng/w1_{nm}"], out_hw)  # noqa: E731
        offs = np.stack([g[f"rng/{hw}1_{nm}"] for nm in ("obs", "next_obs", "pos") for hw in ("h", "w")]).astype(np.int32)
    else:
        crop = lambda src, nm: src[idxs].copy()  # noqa: E731
        offs = np.zeros((6, len(idxs)), dtype=np.int32)
    actor, critic, target, W = synthetic_state(c, out_hw, num_layers=m["num_layers"], hidden=m["hidden"],
                                               seeds=m["weight_seeds"])
    return dict(m=m, obses=obses, nexts=nexts, acts=acts, rews=rews, dones=dones, idxs=idxs, offs=offs,
                obs=crop(obses, "obs"), next_obs=crop(nexts, "next_obs"), pos=crop(obses, "pos"),
                actor=actor, critic=critic, target=target, W=W, log_alpha=torch.tensor(np.log(0.1)))
