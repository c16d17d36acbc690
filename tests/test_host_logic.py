"""Host-side logic of the drop-in surface, on CPU: construction parity with the
reference (same seed => same initial parameters), state_dict layout, weight
tying, replay ring bookkeeping and RNG order, the kernel launch schedule of
update() (via the trace hook: nothing is computed), error behaviour."""
import collections

import numpy as np
import pytest
import torch

import curla_amd
from curla_amd import _lib
from oracle import curla_oracle as O
from tests._util import assert_close, load, sub, summarize

HP = dict(discount=0.99, init_temperature=0.1, alpha_lr=1e-4, alpha_beta=0.5, actor_lr=1e-3, actor_beta=0.9,
          actor_log_std_min=-10, actor_log_std_max=2, actor_update_freq=2, critic_lr=1e-3, critic_beta=0.9,
          critic_tau=0.01, critic_target_update_freq=2, encoder_feature_dim=50, encoder_lr=1e-3, encoder_tau=0.05,
          num_layers=4, num_filters=32, log_interval=1)


class NullLogger:
    def log(self, *a, **k):
        pass


def tiny_agent(**kw):
    aug = curla_amd.RandomCrop((34, 40), (28, 34))
    curla_amd.set_seed_everywhere(1)
    return curla_amd.CurlSacAgent((9, 28, 34), (2,), "cpu", aug, hidden_dim=64, **{**HP, **kw}), aug


def test_construction_matches_reference_init():
    """Same seed, same construction order, same init functions => the reference's
    initial parameters (curl_sac.py:38-54,226-318), bit for bit in summary."""
    g = load("init_tiny.npz")
    agent, _ = tiny_agent()
    for name, mod in (("actor", agent.actor), ("critic", agent.critic), ("critic_target", agent.critic_target)):
        sd = mod.state_dict()
        ref = sub(g, name + "/", as_torch=False)
        assert list(sd.keys()) == list(ref.keys())
        for k, v in ref.items():
            assert_close(summarize(sd[k]), v, 1e-6, f"init {name}/{k}")
    assert_close(summarize(agent.CURL.W), g["W"], 1e-6, "init W")
    assert agent.log_alpha.dtype == torch.float64 and float(agent.log_alpha) == float(g["log_alpha"])
    assert agent.target_entropy == -2


def test_parameter_sharing_and_optimizer_groups():
    agent, _ = tiny_agent()
    for i in range(4):  # curl_sac.py:290 / encoder.py:112-116
        assert agent.actor.encoder.convs[i].weight is agent.critic.encoder.convs[i].weight
        assert agent.actor.encoder.convs[i].bias is agent.critic.encoder.convs[i].bias
    assert agent.actor.encoder.fc.weight is not agent.critic.encoder.fc.weight
    assert agent.CURL.encoder is agent.critic.encoder and agent.CURL.encoder_target is agent.critic_target.encoder
    n = lambda opt: sum(len(g["params"]) for g in opt.param_groups)  # noqa: E731
    # tensors that ever receive a gradient in each optimizer (18/24/1/12/25 minus the
    # never-stepped tied convs (8) and target-encoder tensors (12))
    assert (n(agent.actor_optimizer), n(agent.critic_optimizer), n(agent.log_alpha_optimizer),
            n(agent.encoder_optimizer), n(agent.cpc_optimizer)) == (10, 24, 1, 12, 13)
    assert agent.critic_optimizer.param_groups[0]["betas"] == (0.9, 0.999)
    assert agent.log_alpha_optimizer.param_groups[0]["betas"] == (0.5, 0.999)
    assert curla_amd.PixelEncoder is curla_amd.CNNEncoder


def test_flat_layout_and_twin_stride():
    agent, _ = tiny_agent()
    lay = agent._lay
    flat, gflat = agent._critic_flat, agent._critic_gflat
    assert flat.numel() == lay["total"] == gflat.numel()
    q1, q2 = agent.critic.Q1.trunk, agent.critic.Q2.trunk
    for j in (0, 2, 4):
        for a, b in ((q1[j].weight, q2[j].weight), (q1[j].bias, q2[j].bias)):
            assert (b.data_ptr() - a.data_ptr()) // 4 == agent.critic.twin_stride
            assert a.data_ptr() % 16 == 0 and a.grad.data_ptr() % 16 == 0
    # parameters are views into the flat buffer; grads into its mirror
    w = agent.critic.encoder.convs[2].weight
    off = (w.data_ptr() - flat.data_ptr()) // 4
    assert lay["enc"][0] <= off < lay["enc"][1]
    assert (w.grad.data_ptr() - gflat.data_ptr()) // 4 == off
    assert (agent.CURL.W.data_ptr() - flat.data_ptr()) == 0
    # target has the same layout
    tw = agent.critic_target.encoder.convs[2].weight
    assert (tw.data_ptr() - agent._target_flat.data_ptr()) // 4 == off


def test_fc_weight_checkpoint_layout_roundtrip():
    """fc.weight is held (y,x,c)-ordered; state_dict()/load_state_dict() speak the
    reference's (c,y,x) order (encoder.py:89 flatten of NCHW)."""
    agent, _ = tiny_agent()
    enc = agent.critic.encoder
    c, h, w = 32, *enc.out_dim
    ref_w = torch.arange(50 * c * h * w, dtype=torch.float32).reshape(50, c * h * w)
    sd = agent.critic.state_dict()
    sd["encoder.fc.weight"] = ref_w.clone()
    agent.critic.load_state_dict(sd)
    internal = enc.fc.weight.detach()
    # element (f, c, y, x) of the reference weight sits at column (y*W + x)*C + c internally
    f_, c_, y_, x_ = 7, 5, 3, 2
    assert internal[f_, (y_ * w + x_) * c + c_] == ref_w[f_, (c_ * h + y_) * w + x_]
    assert torch.equal(agent.critic.state_dict()["encoder.fc.weight"], ref_w)
    agent.critic_target.load_state_dict(agent.critic.state_dict())
    assert torch.equal(agent.critic_target.encoder.fc.weight, enc.fc.weight)


def test_encoder_shapes_and_errors():
    for (h, w), L, want in (((76, 135), 4, (31, 61)), ((90, 160), 4, (38, 73)), ((84, 84), 4, (35, 35)),
                            ((76, 76), 4, (31, 31)), ((168, 168), 6, (73, 73)), ((64, 64), 2, (29, 29))):
        e = curla_amd.CNNEncoder((9, h, w), 50, num_layers=L)
        assert tuple(e.out_dim) == want == O.conv_out_hw(h, w, L)
        assert e.fc.in_features == 32 * want[0] * want[1]
    with pytest.raises(NotImplementedError):  # encoder.py:46-47
        curla_amd.CNNEncoder((9, 8, 8), 50, num_layers=4)
    with pytest.raises(ValueError):  # augmentations.py:220
        curla_amd.make_augmentor("nope", (84, 84))
    assert curla_amd.make_augmentor("random_crop", (84, 84)).output_shape == (71, 71)  # augmentations.py:23-24
    assert curla_amd.make_augmentor("random_crop", (90, 160)).output_shape == (76, 135)
    assert curla_amd.make_augmentor("identity", (84, 84)).output_shape == (84, 84)


def test_random_crop_host_path_bit_exact():
    g = load("crop84.npz")
    aug = curla_amd.RandomCrop((84, 84), (76, 76))
    imgs = np.random.RandomState(int(g["imgs_seed"])).randint(0, 256, (16, 9, 84, 84), dtype=np.uint8)
    np.random.seed(int(g["numpy_seed"]))
    out = aug.training_augmentation(imgs)
    assert np.array_equal(out[:2], g["out_first2"])
    import hashlib
    assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == str(g["out_sha256"])
    assert np.array_equal(aug.evaluation_augmentation(imgs[0]), g["center_crop0"])


def test_replay_ring_bookkeeping_and_rng_order():
    aug = curla_amd.RandomCrop((34, 40), (28, 34))
    rb = curla_amd.ReplayBuffer((9, 34, 40), (2,), 5, 8, "cpu", aug)
    rs = np.random.RandomState(0)
    frames = rs.randint(0, 256, (7, 9, 34, 40), dtype=np.uint8)
    for i in range(7):  # wraps: capacity 5
        assert rb.full == (i >= 5)
        rb.add(frames[i], [0.1 * i, -0.1 * i], float(i), frames[(i + 1) % 7], i == 3)
    assert rb.idx == 2 and rb.full and len(rb) == 5
    # slot 0 and 1 were overwritten by transitions 5 and 6; frames are stored HWC
    assert np.array_equal(rb.obses[1].numpy(), frames[6].transpose(1, 2, 0))
    assert np.array_equal(rb.next_obses[4].numpy(), frames[5].transpose(1, 2, 0))
    assert float(rb.rewards[0]) == 5.0 and float(rb.not_dones[3]) == 0.0 and float(rb.not_dones[4]) == 1.0
    assert np.allclose(rb.actions[1].numpy(), [0.6, -0.6])
    # RNG order == the reference's (golden stream replay, utils.py:147 + augmentations.py:66-67 x3)
    g = load("tiny.npz")
    rb2 = curla_amd.ReplayBuffer((9, 34, 40), (2,), 64, 8, "cpu", aug)
    rb2.idx = int(g["meta/n_valid"])
    st = list(np.random.get_state())
    st[1], st[2] = g["meta/numpy_state_keys"], int(g["meta/numpy_state_pos"])
    np.random.set_state(tuple(st))
    idxs, offs = rb2.draw_indices()
    assert np.array_equal(idxs, g["rng/idxs"])
    for j, nm in enumerate(("obs", "next_obs", "pos")):
        assert np.array_equal(offs[2 * j], g[f"rng/h1_{nm}"]) and np.array_equal(offs[2 * j + 1], g[f"rng/w1_{nm}"])
    # identity augmentation draws only the indices (utils.py:168-182)
    rb3 = curla_amd.ReplayBuffer((9, 34, 40), (2,), 64, 8, "cpu", curla_amd.IdentityAugmentation((34, 40)))
    rb3.idx = 40
    np.random.seed(3)
    a, offs3 = rb3.draw_indices()
    np.random.seed(3)
    assert np.array_equal(a, np.random.randint(0, 40, size=8)) and not offs3.any()


def test_sampling_pixels_without_gpu_fails_loudly():
    aug = curla_amd.RandomCrop((34, 40), (28, 34))
    rb = curla_amd.ReplayBuffer((9, 34, 40), (2,), 8, 4, "cpu", aug)
    rb.add(np.zeros((9, 34, 40), np.uint8), [0, 0], 0.0, np.zeros((9, 34, 40), np.uint8), False)
    with pytest.raises(RuntimeError):
        rb.sample_cpc()
    agent, _ = tiny_agent()
    with pytest.raises(RuntimeError):
        agent.update(rb, NullLogger(), 0)


def _trace_updates(agent, rb, steps, **kw):
    calls = []
    _lib.set_trace_hook(lambda n, a: calls.append(n))
    try:
        out = []
        for s in steps:
            calls.clear()
            agent.update(rb, NullLogger(), s, **kw)
            out.append(collections.Counter(calls))
        return out
    finally:
        _lib.set_trace_hook(None)


def _filled_rb(aug, hw=(34, 40)):
    rb = curla_amd.ReplayBuffer((9,) + hw, (2,), 32, 8, "cpu", aug)
    for i in range(12):
        rb.add(np.zeros((9,) + hw, np.uint8), [0, 0], 0.0, np.zeros((9,) + hw, np.uint8), False)
    return rb


def test_update_schedule_launch_counts():
    """curl_sac.py:426-451 schedule, with the conv passes the build shares:
    5 conv-stack forwards + 2 backwards on every step, as TWO launch sequences: the critic phase's
    [obs | next_obs] through the online convs (one minibatch of 2B, ObsRef.pair) together with next_obs through the
    target convs (second problem of the same launches), and obs through the stepped convs together with the positives
    through the target convs (actor phase on even steps, CURL phase on odd ones)."""
    agent, aug = tiny_agent()
    even, odd = _trace_updates(agent, _filled_rb(aug), [0, 1])
    for c in (even, odd):
        assert c["curla_conv1_fwd2"] == 2 and c["curla_conv3x3_s1_fwd2"] == 6
        assert c["curla_conv1_fwd"] == 0 and c["curla_conv3x3_s1_fwd"] == 0
        # per backward pass: one launch per stride-1 layer (weight + data gradient together), the first layer's weight
        # gradient, and one slab reduction for all layers
        assert c["curla_conv1_wgrad_slabs"] == 2 and c["curla_conv3x3_s1_bwd_slabs"] == 6
        assert c["curla_conv3x3_s1_dgrad"] == 0 and c["curla_wgrad_reduce_multi"] == 2
        # the TD loss (and, actor steps, the actor / alpha loss) is evaluated inside the backward launch of the Q
        # functions' last layer
        assert c["curla_curl_ce"] == 1 and c["curla_critic_td_loss"] == 0 and c["curla_actor_loss"] == 0
    assert even["curla_mlp_out_bwd_loss"] == 2 and odd["curla_mlp_out_bwd_loss"] == 1  # actor_update_freq = 2
    assert even["curla_soft_update2"] == 1 and odd["curla_soft_update2"] == 0   # critic_target_update_freq = 2
    # launches that are neither convolutions nor dense layers (GEMMs / last-layer kernels): LayerNorm pieces, policy
    # head, losses, bias-gradient sums, the scalar gather, the target lerp -- kept to about twenty per even update
    dense = ("curla_conv", "curla_gemm", "curla_mlp_out", "curla_fc_", "curla_linear_bwd")
    small = {k: v for k, v in even.items() if not k.startswith(dense)}
    assert sum(small.values()) <= 13, small
    # what used to be launches of their own and now rides in another: the LayerNorms of the encoders of a phase (one
    # launch for three / two of them), the policy head (inside the actor trunk's last-layer launch), the four Q
    # functions of target + critic (one two-level batch per layer)
    assert even["curla_fc_ln_fwd_multi"] == 2 and odd["curla_fc_ln_fwd_multi"] == 2 and even["curla_fc_ln_fwd"] == 0
    assert even["curla_mlp_out_head_fwd"] == 2 and odd["curla_mlp_out_head_fwd"] == 1 and even["curla_actor_head_fwd"] == 0
    assert even["curla_gemm_nested"] == 2 and even["curla_mlp_out_fwd_nested"] == 1
    assert sum(even.values()) <= 65 and sum(odd.values()) <= 46
    # a linear layer's weight and data gradient share a launch: hidden + first layer of the twin Q functions (critic
    # phase) and of the actor trunk (actor phase; the Q functions there only pass the gradient down: curla_gemm)
    assert even["curla_linear_bwd"] == 4 and odd["curla_linear_bwd"] == 2
    # fc backward: data + weight gradient in one launch where the conv stack gets a gradient (critic, CURL), the
    # weight gradient alone in the actor phase (encoder detached)
    # -- and each of them finishes the LayerNorm / fc-bias gradients its LayerNorm backward left as partial sums
    # (curla_ln_bwd_partial): no parameter-gradient launch of their own
    assert even["curla_fc_bwd_ln"] == 2 and even["curla_fc_dw_ln"] == 1 and even["curla_fc_dx"] == 0
    assert even["curla_fc_bwd"] == 0 and even["curla_fc_dw"] == 0
    assert even["curla_ln_bwd_partial"] == 3 and even["curla_ln_bwd"] == 0 and even["curla_ln_bwd_twin"] == 0
    assert even["curla_split_sum"] == 0
    assert even["curla_gemm"] <= 40 and even["curla_concat"] == 0 and even["curla_td_target"] == 0
    # only_cpc (train.py:425): no SAC phases
    (c,) = _trace_updates(agent, _filled_rb(aug), [2], only_cpc=True)
    assert c["curla_critic_td_loss"] == 0 and c["curla_mlp_out_bwd_loss"] == 0 and c["curla_curl_ce"] == 1
    assert c["curla_conv1_fwd2"] == 1 and c["curla_conv1_fwd"] == 0 and c["curla_conv1_wgrad_slabs"] == 1


def test_update_schedule_pixel_sac():
    """--pixel_sac forces identity augmentation and skips CURL (train.py:262-264, curl_sac.py:448)."""
    aug = curla_amd.IdentityAugmentation((34, 40))
    curla_amd.set_seed_everywhere(1)
    agent = curla_amd.CurlSacAgent((9, 34, 40), (2,), "cpu", aug, hidden_dim=64, pixel_sac=True, **HP)
    even, odd = _trace_updates(agent, _filled_rb(aug), [0, 1])
    # critic phase: one two-problem launch ([obs | next_obs] online, next_obs target); actor phase (even): obs alone
    assert even["curla_conv1_fwd2"] == 1 and even["curla_conv1_fwd"] == 1
    assert odd["curla_conv1_fwd2"] == 1 and odd["curla_conv1_fwd"] == 0
    assert even["curla_conv1_wgrad_slabs"] == 1 and odd["curla_conv1_wgrad_slabs"] == 1
    assert even["curla_curl_ce"] == 0


def test_reference_style_buffer_goes_through_tensor_contract():
    """A buffer that only offers the reference's sample_cpc() (float NCHW tensors)
    drives the same phases through the f32 loader."""
    agent, aug = tiny_agent()

    class RefStyle:
        def sample_cpc(self):
            o = torch.zeros(8, 9, 28, 34)
            return o, torch.zeros(8, 2), torch.zeros(8, 1), o.clone(), torch.ones(8, 1), dict(obs_anchor=o, obs_pos=o.clone())
    seen = []
    _lib.set_trace_hook(lambda n, a: seen.append((n, a)))
    try:
        agent.update(RefStyle(), NullLogger(), 0)
    finally:
        _lib.set_trace_hook(None)
    conv1 = [a for n, a in seen if n == "curla_conv1_fwd"]
    assert len(conv1) == 5 and all(a[1] == 0 for a in conv1)  # src_is_u8 == 0


def test_full_checkpoint_file_layout(tmp_path):
    """save_checkpoint writes reference-layout tensors (+ optimizer, RNG, step) and load_checkpoint refuses other files."""
    import curla_amd
    aug = curla_amd.RandomCrop((40, 40), (34, 34))
    a = curla_amd.CurlSacAgent((9, 34, 34), (2,), torch.device("cpu"), aug, hidden_dim=32)
    p = tmp_path / "ck.pt"
    a.save_checkpoint(str(p), 11)
    ck = torch.load(p, weights_only=False)
    assert ck["format"] == "curla_amd.checkpoint.v1" and ck["step"] == 11
    assert set(ck["optimizers"]) == {"actor", "critic", "log_alpha", "encoder", "cpc"}
    assert list(ck["critic"].keys()) == list(a.critic.state_dict().keys())
    b = curla_amd.CurlSacAgent((9, 34, 34), (2,), torch.device("cpu"), aug, hidden_dim=32)
    assert b.load_checkpoint(str(p)) == 11
    for k, v in a.actor.state_dict().items():
        assert torch.equal(v, b.actor.state_dict()[k]), k
    torch.save({"format": "x"}, tmp_path / "bad.pt")
    with pytest.raises(ValueError):
        b.load_checkpoint(str(tmp_path / "bad.pt"))


class _FakeEnv:
    """Minimal environment for FrameStack: random RGB frames, fixed-length episodes."""

    def __init__(self, hw, episode_len, seed):
        self.rs = np.random.RandomState(seed)
        self.hw, self.episode_len, self.t = hw, episode_len, 0
        self.observation_space = type("S", (), dict(shape=(3,) + hw, dtype=np.uint8))()
        self._max_episode_steps = episode_len
        self.curl_driving = False

    def _frame(self):
        return self.rs.randint(0, 256, (3,) + self.hw, dtype=np.uint8)

    def reset(self):
        self.t = 0
        return self._frame()

    def step(self, action):
        self.t += 1
        return self._frame(), 1.0, self.t >= self.episode_len, {}


def _rollout(env, n):
    """n transitions as train.py collects them (train.py:396-445): obs = next_obs inside an episode."""
    out = []
    obs = env.reset()
    for i in range(n):
        nxt, r, done, _ = env.step(None)
        out.append((obs, np.array([0.01 * i, -0.02 * i], np.float32), float(i), nxt, done))
        obs = env.reset() if done else nxt
    return out


def test_frame_stack_wrapper():
    """utils.py:238-268: reset() returns k copies of the first frame, step() shifts the new frame in."""
    env = curla_amd.utils.FrameStack(_FakeEnv((8, 10), 5, 0), 3)
    assert env.observation_space.shape == (9, 8, 10) and env._max_episode_steps == 5
    o0 = env.reset()
    assert o0.shape == (9, 8, 10) and np.array_equal(o0[:3], o0[3:6]) and np.array_equal(o0[:3], o0[6:])
    o1, r, done, info = env.step(None)
    assert np.array_equal(o1[:6], o0[3:]) and not np.array_equal(o1[6:], o0[6:]) and r == 1.0 and not done
    assert env.hw == (8, 10)  # other attributes are the wrapped environment's


def test_dedup_frame_store_bookkeeping():
    """dedup_frames=True: one new RGB frame per environment step (two at an episode start), byte-exact stacks,
    frames released when the transition ring wraps, loud failure when the caller's observations share nothing."""
    hw, k, cap = (8, 10), 3, 12
    aug = curla_amd.IdentityAugmentation(hw)
    rb = curla_amd.ReplayBuffer((3 * k,) + hw, (2,), cap, 4, "cpu", aug, dedup_frames=True)
    trans = _rollout(curla_amd.utils.FrameStack(_FakeEnv(hw, 7, 1), k), 30)
    new_frames = []
    for t, (o, a, r, n, d) in enumerate(trans):
        before = rb.frames_in_use()[0]
        rb.add(o, a, r, n, d)
        new_frames.append(rb.frames_in_use()[0] - before)
        live = range(max(0, t + 1 - cap), t + 1)
        # every live transition still reads back byte for byte
        for u in live:
            slot = u % cap
            for which, want in ((0, trans[u][0]), (1, trans[u][3])):
                ids = rb._fid_h[slot, which]
                got = np.concatenate([rb.frames[int(f)].permute(2, 0, 1).numpy() for f in ids])
                assert np.array_equal(got, want), (t, u, which)
        assert float(rb.rewards[t % cap]) == r and float(rb.not_dones[t % cap]) == float(not d)
    # first transition of an episode: 2 new frames (k copies of the first frame + the next one); otherwise 1 --
    # minus what the wrapping ring released
    assert new_frames[0] == 2 and new_frames[1] == 1 and new_frames[6] == 1 and new_frames[7] == 2
    used, total = rb.frames_in_use()
    assert used <= cap + cap // 7 + 2 * k and total == rb.frame_capacity  # ~1 frame per live transition
    assert int((rb._store.refs > 0).sum()) == used and rb.full and rb.idx == 30 % cap
    # observations that share nothing exhaust a store sized for stacked episodes -- loudly
    rb2 = curla_amd.ReplayBuffer((3 * k,) + hw, (2,), 8, 4, "cpu", aug, dedup_frames=True, frame_capacity=20)
    rs = np.random.RandomState(0)
    with pytest.raises(MemoryError):
        for _ in range(8):
            rb2.add(rs.randint(0, 256, (9,) + hw, dtype=np.uint8), [0, 0], 0.0, rs.randint(0, 256, (9,) + hw, dtype=np.uint8), False)
    with pytest.raises(ValueError):
        curla_amd.ReplayBuffer((4,) + hw, (2,), 8, 4, "cpu", aug, dedup_frames=True)


def test_stale_minibatch_handle_raises():
    """Each sample gets its own device index block; a handle older than N_SAMPLE_SLOTS samples is refused
    instead of silently reading another minibatch's pixels."""
    aug = curla_amd.RandomCrop((34, 40), (28, 34))
    rb = _filled_rb(aug)
    _lib.set_trace_hook(lambda n, a: None)
    try:
        first = rb.sample_cpc_refs()[0]
        second = rb.sample_cpc_refs()[0]
        assert first.idx.data_ptr() != second.idx.data_ptr()
        first.check(), second.check()
        rb.sample_cpc_refs()
        second.check()
        with pytest.raises(_lib.CurlaHipError):
            first.check()
    finally:
        _lib.set_trace_hook(None)


def test_augmentor_tensor_api_needs_the_device():
    """ColorJiggle / NoisyCover.training_augmentation run the HIP kernels; a CPU tensor is refused (no fallback)
    instead of being passed through un-augmented."""
    for name in ("color_jiggle", "noisy_cover"):
        aug = curla_amd.make_augmentor(name, (20, 24))
        with pytest.raises(RuntimeError):
            aug.training_augmentation(torch.zeros(2, 9, 20, 24))
        with pytest.raises(ValueError):
            aug.training_augmentation(torch.zeros(2, 9, 20, 25))
    assert curla_amd.make_augmentor("identity", (20, 24)).training_augmentation("x") == "x"


def test_flat_adam_plan_and_state_format():
    """FlatAdam (curla_amd/optim.py) keeps torch.optim.Adam's param_groups / state_dict format and turns a step into
    one launch per contiguous run of live parameters (checked through the launch trace: nothing computes here)."""
    from curla_amd import _lib
    from curla_amd.optim import FlatAdam
    sizes, offs, off = [(8, 5), (8,), (3,), (16, 4), (1,)], [], 0
    for n in sizes:
        offs.append(off)
        off += (int(np.prod(n)) + 3) & ~3
    flat, gflat = torch.zeros(off), torch.zeros(off)
    params = []
    for n, o in zip(sizes, offs):
        p = torch.nn.Parameter(flat[o:o + int(np.prod(n))].view(n))
        p.grad = gflat[o:o + int(np.prod(n))].view(n)
        params.append(p)
    opt = FlatAdam(params, flat, gflat, lr=3e-4, betas=(0.5, 0.999))
    assert isinstance(opt, torch.optim.Adam) and opt.param_groups[0]["lr"] == 3e-4
    calls = []
    _lib.set_trace_hook(lambda name, args: calls.append((name, args)))
    try:
        opt.step()
        assert [c[0] for c in calls] == ["curla_adam_step"]
        assert calls[0][1][4] == off - 3 and calls[0][1][9] == 1  # one run over everything (last slot: 1 + 3 padding)
        calls.clear()
        g2 = params[2].grad
        params[2].grad = None   # a hole in the middle: two runs, and the skipped parameter keeps step 1
        opt.step()
        assert [(c[1][4], c[1][9]) for c in calls] == [(48, 2), (65, 2)]
        calls.clear()
        params[2].grad = g2     # back: its step count differs from its neighbours', so the run is split in three
        opt.step()
        assert [(c[1][4], c[1][9]) for c in calls] == [(48, 3), (3, 2), (65, 3)]
    finally:
        _lib.set_trace_hook(None)
    sd = opt.state_dict()
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(n)) for n in sizes], lr=3e-4, betas=(0.5, 0.999))
    for p in ref.param_groups[0]["params"]:
        p.grad = torch.zeros_like(p)
    ref.step()
    rsd = ref.state_dict()
    assert set(sd["state"]) == set(rsd["state"]) and set(sd["state"][0]) == set(rsd["state"][0])
    assert {k: v for k, v in sd["param_groups"][0].items() if k not in ("params",)}.keys() == \
        {k: v for k, v in rsd["param_groups"][0].items() if k not in ("params",)}.keys()
    assert [float(sd["state"][i]["step"]) for i in range(5)] == [3, 3, 2, 3, 3]
    opt2 = FlatAdam(params, flat, gflat, lr=3e-4, betas=(0.5, 0.999))
    opt2.load_state_dict(sd)
    assert opt2._steps == [3, 3, 2, 3, 3]
    assert opt2.state[params[0]]["exp_avg"].data_ptr() == opt2._m.data_ptr()  # moments are views of the flat mirror


def _rider_cases():
    rs = np.random.RandomState(4)
    vals = [0.0, 1.0, -1.0, 0.1, -0.1, 1e-3, 3.141592653589793, -2.718281828459045e-7, 123456.789, -2.0 ** 27, 2.0 ** 28 - 2.0 ** -20,
            2.0 ** -60, -2.0 ** -79]
    vals += list(rs.randn(64) * 10.0 ** rs.uniform(-8, 6, 64))
    return [float(v) for v in vals]


def test_f64_rider_words_are_exact_and_sum_exactly():
    """log_alpha's float64 gradient inside the actor's float32 all-reduce bucket (ops.f64_words_of /
    f64_of_words are the host statement of curla_f64_pack / curla_f64_unpack): the 8 words are integers below 2^20 that
    reproduce the value exactly; sums of the words over 2 .. 16 ranks (exact in float32) decode to the correctly rounded
    EXACT sum of the ranks' doubles -- for two ranks the float64 all-reduce's (a + b) / 2, bit for bit -- whether the
    collective summed or averaged (power-of-two worlds)."""
    from fractions import Fraction

    from curla_amd import ops
    vals = _rider_cases()
    for v in vals:
        w = ops.f64_words_of(v)
        assert len(w) == ops.F64_WORDS and all(float(np.float32(x)) == x and abs(x) < 2 ** 20 and x == int(x) for x in w)
        assert ops.f64_of_words(w) == v, v
    rs = np.random.RandomState(5)
    for world in (2, 2, 2, 4, 8, 16):
        for _ in range(40):
            picks = [vals[i] for i in rs.randint(0, len(vals), world)]
            words = np.stack([np.array(ops.f64_words_of(v), dtype=np.float32) for v in picks])
            summed = words.sum(0, dtype=np.float32)       # what a SUM all-reduce leaves ...
            assert np.array_equal(summed.astype(np.float64), words.astype(np.float64).sum(0))  # ... exactly
            averaged = (summed / np.float32(world))       # ... and ncclAvg (exact: power of two)
            exact = sum(Fraction(v) for v in picks)
            want = float(exact) / world                   # one rounding of the exact sum, then the exact division
            assert ops.f64_of_words(summed.tolist(), 1, world) == want
            assert ops.f64_of_words(averaged.tolist(), world, world) == want
            if world == 2:
                assert want == (np.float64(picks[0]) + np.float64(picks[1])) / np.float64(2)
    # below the last digit (2^-132) a value is dropped: an absolute error no gradient step can see
    assert ops.f64_of_words(ops.f64_words_of(1e-50)) == 0.0 and abs(ops.f64_of_words(ops.f64_words_of(1e-30)) - 1e-30) < 2.0 ** -132
    # out of range / non-finite values travel in word 0 and come back finite-or-not as they went
    assert ops.f64_of_words(ops.f64_words_of(2.0 ** 30)) == 2.0 ** 30
    assert np.isnan(ops.f64_of_words(ops.f64_words_of(float("nan"))))
    assert np.isinf(ops.f64_of_words(ops.f64_words_of(float("inf"))))
    # a world size that is not a power of two: averaged digits are rounded to 24 bits -- 2^-24 of the value
    picks = [vals[3], vals[5], vals[6]]
    words = np.stack([np.array(ops.f64_words_of(v), dtype=np.float32) for v in picks]).sum(0, dtype=np.float32) / np.float32(3)
    got = ops.f64_of_words(words.tolist(), 3, 3)
    assert abs(got - sum(picks) / 3) <= 2.0 ** -22 * abs(sum(picks) / 3)
