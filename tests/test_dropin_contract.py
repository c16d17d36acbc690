"""The drop-in boundary as the reference's drivers use it.

train.py / eval.py reach the learner path through ``utils.`` (train.py:146,247,260,286-290,317,416; eval.py:77,121,142),
``make_augmentor`` (train.py:21,268), ``CurlSacAgent(...)`` with the keyword list of train.py:199-227, and the
per-step calls of train.py:408-443 (``sample_action`` under ``eval_mode``, ``update(replay_buffer, L, step
[, only_cpc=True])``, ``replay_buffer.add``), train.py:353-370 (``select_action`` in the evaluation loop,
``agent.save``, ``replay_buffer.save``) and eval.py:165 (``agent.load``).

The CPU part applies each binding INTEGRATION.md documents and checks every attribute and signature those lines touch,
then walks the loop's call sequence on the launch-trace hook (host logic only, nothing is computed).  The ``-m gpu``
part drives the same sequence for 50 environment steps on the device with a stand-in environment (own code: the
reference's CarlaEnv needs the simulator) and checks what can be checked without the simulator: stored transitions
are the bytes that were added, the logged keys are the reference's, checkpoints round-trip into a fresh agent."""
import importlib
import inspect
import os
import sys

import numpy as np
import pytest
import torch

# keyword list of make_agent (train.py:199-227) with the values of train.py's argparse defaults (train.py:73-104)
TRAIN_PY_AGENT_KWARGS = dict(
    hidden_dim=1024, discount=0.99, init_temperature=0.1, alpha_lr=1e-4, alpha_beta=0.5, actor_lr=1e-3, actor_beta=0.9,
    actor_log_std_min=-10, actor_log_std_max=2, actor_update_freq=2, critic_lr=1e-3, critic_beta=0.9, critic_tau=0.01,
    critic_target_update_freq=2, encoder_feature_dim=50, encoder_lr=1e-3, encoder_tau=0.05, num_layers=4, num_filters=32,
    log_interval=100, log_param_hist_imgs=False, detach_encoder=False, pixel_sac=False)

# the keys CurlSacAgent hands its logger (curl_sac.py:361,381-386,400-401,422,431)
AGENT_LOG_KEYS = {"train/batch_reward", "train_critic/loss", "train_actor/loss", "train_actor/target_entropy",
                  "train_actor/entropy", "train_alpha/loss", "train_alpha/value", "train/curl_loss"}


class _Space:
    def __init__(self, shape, dtype, rng=None):
        self.shape, self.dtype, self._rng = tuple(shape), dtype, rng

    def sample(self):
        return self._rng.uniform(-1.0, 1.0, self.shape).astype(np.float32)


class StandInEnv:
    """What train.py needs of an environment (carla_env.py:489-499 spaces; train.py:342-343,432 attributes): uint8
    (3, H, W) camera frames, a 2-vector action, episodes of ``_max_episode_steps`` steps."""

    def __init__(self, hw, episode_steps, seed):
        self._rng = np.random.RandomState(seed)
        self.observation_space = _Space((3,) + tuple(hw), np.uint8)
        self.action_space = _Space((2,), np.float32, self._rng)
        self.fps, self.dt, self.desired_speed, self.verbose = 20, 0.05, 65.0, False
        self._max_episode_steps = episode_steps
        self.curl_driving = False
        self._t = 0
        self.deactivated = False
        self.frames = []          # every frame handed out, in order
        self.driving_seen = []    # the curl_driving flag the wrapper forwarded at each step (utils.py:262)

    def _frame(self):
        f = self._rng.randint(0, 256, self.observation_space.shape, dtype=np.uint8)
        self.frames.append(f)
        return f

    def reset(self):
        self._t = 0
        self.curl_driving = False
        return self._frame()

    def step(self, action):
        assert np.asarray(action).shape == (2,)
        self.driving_seen.append(self.curl_driving)
        self._t += 1
        done = self._t >= self._max_episode_steps
        info = {k: float(self._t) for k in ("r1", "r2", "r3", "r4", "r5", "mean_kmh", "max_kmh", "brake_sum")}
        return self._frame(), float(self._rng.randn()), done, info

    def deactivate(self):
        self.deactivated = True


class RecordingLogger:
    """The Logger duck type the agent and the loop use (logger.py:139-177): log / dump (+ the histogram hooks)."""

    def __init__(self):
        self.values, self.dumps = {}, []

    def log(self, key, value, step, n=1):
        self.values.setdefault(key, []).append((step, value))

    def dump(self, step):
        self.dumps.append(step)

    def log_histogram(self, *a, **k):
        pass

    def log_param(self, *a, **k):
        pass

    def log_image(self, *a, **k):
        pass


def _bind(how):
    """The three bindings INTEGRATION.md section 1 documents; returns (utils, make_augmentor, CurlSacAgent)."""
    if how == "submodules":      # import curla_amd.utils as utils; from curla_amd.augmentations import ...
        import curla_amd.utils as utils
        from curla_amd.augmentations import make_augmentor
        from curla_amd.curl_sac import CurlSacAgent
    elif how == "package":       # import curla_amd as utils; from curla_amd import ...
        import curla_amd as utils
        from curla_amd import CurlSacAgent, make_augmentor
    else:                        # dropin.install(): the reference's own import lines, unedited (train.py:20-25)
        import curla_amd.dropin as dropin
        dropin.install()
        try:
            utils = importlib.import_module("utils")
            make_augmentor = importlib.import_module("augmentations").make_augmentor
            CurlSacAgent = importlib.import_module("curl_sac").CurlSacAgent
            assert importlib.import_module("encoder").PixelEncoder is importlib.import_module("encoder").CNNEncoder
        finally:
            dropin.uninstall()
        assert "utils" not in sys.modules and "curl_sac" not in sys.modules
    return utils, make_augmentor, CurlSacAgent


def _params(fn):
    return list(inspect.signature(fn).parameters)


@pytest.mark.parametrize("how", ["submodules", "package", "dropin"])
def test_documented_bindings_resolve_every_name_the_drivers_touch(how):
    utils, make_augmentor, CurlSacAgent = _bind(how)
    # train.py:146,247,260,286-290,317,416 / eval.py:77,121,142 / curl_sac.py:443-445
    for name in ("eval_mode", "FrameStack", "set_seed_everywhere", "make_dir", "ReplayBuffer", "soft_update_params",
                 "module_hash", "preprocess_obs"):
        assert callable(getattr(utils, name)), name
    # train.py:317-324: keyword call
    assert _params(utils.ReplayBuffer.__init__)[1:7] == ["obs_shape", "action_shape", "capacity", "batch_size", "device",
                                                          "augmentor"]
    assert _params(utils.ReplayBuffer.add)[1:] == ["obs", "action", "reward", "next_obs", "done"]      # train.py:443
    assert _params(utils.ReplayBuffer.save)[1:] == ["save_dir"] == _params(utils.ReplayBuffer.load)[1:]  # train.py:370
    assert _params(utils.FrameStack.__init__)[1:] == ["env", "k"]                                        # train.py:247
    assert _params(utils.make_dir) == ["dir_path"] and _params(utils.set_seed_everywhere) == ["seed"]
    # train.py:199-227: every keyword make_agent passes is accepted, in the reference's order (curl_sac.py:226-256)
    ctor = _params(CurlSacAgent.__init__)[1:]
    assert ctor[:4] == ["obs_shape", "action_shape", "device", "augmentor"]
    assert set(TRAIN_PY_AGENT_KWARGS) <= set(ctor)
    assert ctor == ["obs_shape", "action_shape", "device", "augmentor", "hidden_dim", "discount", "init_temperature",
                    "alpha_lr", "alpha_beta", "actor_lr", "actor_beta", "actor_log_std_min", "actor_log_std_max",
                    "actor_update_freq", "critic_lr", "critic_beta", "critic_tau", "critic_target_update_freq",
                    "encoder_feature_dim", "encoder_lr", "encoder_tau", "num_layers", "num_filters", "cpc_update_freq",
                    "log_interval", "log_param_hist_imgs", "detach_encoder", "pixel_sac"]
    # train.py:425-429, 149-151, 418, 368; eval.py:78,165
    assert _params(CurlSacAgent.update)[1:5] == ["replay_buffer", "L", "step", "only_cpc"]  # (+ keyword-only-in-practice test hooks behind)
    assert inspect.signature(CurlSacAgent.update).parameters["only_cpc"].default is False
    assert _params(CurlSacAgent.update_critic)[1:8] == ["obs", "action", "reward", "next_obs", "not_done", "L", "step"]
    assert _params(CurlSacAgent.update_actor_and_alpha)[1:4] == ["obs", "L", "step"]
    assert _params(CurlSacAgent.update_cpc)[1:] == ["obs_anchor", "obs_pos", "cpc_kwargs", "L", "step"]
    assert _params(CurlSacAgent.select_action)[1:] == ["obs"] and _params(CurlSacAgent.sample_action)[1] == "obs"
    assert _params(CurlSacAgent.save)[1:] == ["model_dir", "augmentation", "step"] == _params(CurlSacAgent.load)[1:]
    assert _params(CurlSacAgent.train)[1:] == ["training"]
    # train.py:268-272 and the five README modes (augmentations.py:208-221)
    for name, out in (("identity", (90, 160)), ("random_crop", (76, 135)), ("color_jiggle", (90, 160)),
                      ("noisy_cover", (90, 160))):
        aug = make_augmentor(name, (90, 160))
        assert tuple(aug.output_shape) == out
        assert callable(aug.training_augmentation) and callable(aug.evaluation_augmentation)


def test_dropin_install_refuses_to_shadow_half_of_the_path():
    import types

    import curla_amd.dropin as dropin
    sys.modules["utils"] = types.ModuleType("utils")  # somebody else's utils imported first
    try:
        with pytest.raises(ImportError):
            dropin.install()
        assert "curl_sac" not in sys.modules
        dropin.install(force=True)
        assert sys.modules["utils"].__name__ == "curla_amd.utils"
    finally:
        dropin.uninstall()
        sys.modules.pop("utils", None)


def _run_training_loop(utils, make_augmentor, CurlSacAgent, device, work_dir, *, camera_hw, batch_size, hidden_dim,
                       num_train_steps, init_steps, eval_freq, episode_steps, acc_steps, log_interval, seed=7,
                       augmentation="random_crop"):
    """train.py:251-457 in this test's own words: the same calls in the same order with the simulator, the video
    recorder and the bookkeeping that does not touch the learner path left out.  Returns what the checks need."""
    utils.set_seed_everywhere(seed)                                             # train.py:260
    augmentor = make_augmentor(augmentation, camera_hw)                         # train.py:268
    aug_hw = tuple(augmentor.output_shape)                                      # train.py:271-272
    dirs = [utils.make_dir(os.path.join(work_dir, d)) for d in ("", "video", "model", "buffer")]  # train.py:286-290
    assert all(os.path.isdir(d) for d in dirs)
    model_dir, buffer_dir = dirs[2], dirs[3]
    L = RecordingLogger()
    frame_stack = 3
    raw = StandInEnv(camera_hw, episode_steps, seed)
    raw.reset()                                                                 # train.py:244
    env = utils.FrameStack(raw, k=frame_stack)                                  # train.py:247
    action_shape = env.action_space.shape                                       # train.py:310
    pre_aug_obs_shape = env.observation_space.shape                             # train.py:313
    assert pre_aug_obs_shape == (3 * frame_stack,) + tuple(camera_hw)
    obs_shape = (3 * frame_stack,) + aug_hw                                     # train.py:314
    replay_buffer = utils.ReplayBuffer(obs_shape=pre_aug_obs_shape, action_shape=action_shape, capacity=64,
                                       batch_size=batch_size, device=device, augmentor=augmentor)  # train.py:317-324
    kwargs = dict(TRAIN_PY_AGENT_KWARGS, hidden_dim=hidden_dim, log_interval=log_interval)
    agent = CurlSacAgent(obs_shape=obs_shape, action_shape=action_shape, device=device, augmentor=augmentor, **kwargs)
    assert env.desired_speed / 3.6 * env.dt * env._max_episode_steps > 0        # train.py:342 (forwarded attributes)

    added, actions_taken, evals = [], [], []
    episode, done, episode_step, obs = 0, True, 0, None
    for step in range(num_train_steps + 1):                                     # train.py:346
        if step % eval_freq == 0:                                               # train.py:353-370
            e_obs, e_done = env.reset(), False
            while not e_done:
                e_obs = augmentor.evaluation_augmentation(e_obs)                # train.py:140
                with utils.eval_mode(agent):                                    # train.py:146
                    assert agent.training is False
                    env.curl_driving = True
                    a = agent.select_action(e_obs)                              # train.py:151
                assert agent.training is True
                assert a.shape == (2,) and a.dtype == np.float32
                e_obs, _, e_done, _ = env.step(a)
            evals.append(step)
            done = True
            agent.save(model_dir, augmentation, step)                           # train.py:368
            replay_buffer.save(buffer_dir)                                      # train.py:370
        if done:                                                                # train.py:373-404
            L.dump(step)
            obs, done, episode_step = env.reset(), False, 0
            episode += 1
        if step < init_steps:                                                   # train.py:408-418
            action = env.action_space.sample()
        elif episode_step < acc_steps:
            action = np.array([0.5, 0.0])
        else:
            with utils.eval_mode(agent):
                env.curl_driving = True
                action = agent.sample_action(obs)      # the un-cropped frame stack: centre-cropped inside
            assert action.shape == (2,)
        if step >= init_steps:                                                  # train.py:421-429
            if episode_step < acc_steps:
                agent.update(replay_buffer, L, step, only_cpc=True)
            else:
                agent.update(replay_buffer, L, step)
        next_obs, reward, done, info = env.step(action)                         # train.py:432
        done_bool = 0 if episode_step + 1 == env._max_episode_steps else float(done)  # train.py:439
        replay_buffer.add(obs, action, reward, next_obs, done_bool)             # train.py:443
        added.append((obs, np.asarray(action, np.float32), np.float32(reward), next_obs, float(not done_bool)))
        actions_taken.append(action)
        obs = next_obs
        episode_step += 1
    env.deactivate()                                                            # train.py:457
    assert raw.deactivated
    return dict(agent=agent, buffer=replay_buffer, L=L, added=added, env=raw, model_dir=model_dir,
                buffer_dir=buffer_dir, augmentor=augmentor, obs_shape=obs_shape, pre_aug_obs_shape=pre_aug_obs_shape,
                evals=evals, kwargs=kwargs, episodes=episode)


LOOP = dict(camera_hw=(44, 60), batch_size=8, hidden_dim=64, num_train_steps=50, init_steps=12, eval_freq=25,
            episode_steps=9, acc_steps=2, log_interval=5)


def test_training_loop_call_sequence_host_side(tmp_path):
    """The loop on the CPU with the launch-trace hook: every call of train.py's sequence is accepted, ring
    bookkeeping advances as the reference's, the wrapper forwards what train.py reads and writes."""
    from curla_amd import _lib
    utils, make_augmentor, CurlSacAgent = _bind("dropin")
    launches = []
    _lib.set_trace_hook(lambda name, args: launches.append(name))
    try:
        r = _run_training_loop(utils, make_augmentor, CurlSacAgent, torch.device("cpu"), str(tmp_path), **LOOP)
    finally:
        _lib.set_trace_hook(None)
    rb = r["buffer"]
    n = LOOP["num_train_steps"] + 1
    assert rb.idx == n % 64 and rb.full is (n >= 64) and len(rb) == 64
    assert r["evals"] == [0, 25, 50]
    assert any(r["env"].driving_seen)  # env.curl_driving = True on the wrapper reaches the wrapped env (utils.py:262)
    assert "curla_conv1_fwd" in launches or "curla_conv1_fwd2" in launches
    assert sorted(os.listdir(r["model_dir"])) == sorted(
        f"random_crop_{kind}_{s}.pt" for s in (0, 25, 50) for kind in ("actor", "critic", "curl"))
    # buffer chunks {start}_{end}.pt continue one another (utils.py:189-202)
    spans = sorted(tuple(int(x) for x in f[:-3].split("_")) for f in os.listdir(r["buffer_dir"]))
    assert spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] == 50
    assert AGENT_LOG_KEYS <= set(r["L"].values)


@pytest.mark.gpu
def test_training_loop_call_sequence_on_the_device(tmp_path):
    """50 environment steps of the train.py sequence on the MI355X through the documented binding."""
    from curla_amd import _lib
    utils, make_augmentor, CurlSacAgent = _bind("dropin")
    dev = torch.device("cuda")
    r = _run_training_loop(utils, make_augmentor, CurlSacAgent, dev, str(tmp_path), **LOOP)
    torch.cuda.synchronize()
    assert _lib._lib is not None, "the native library is what must have run"
    agent, rb, L = r["agent"], r["buffer"], r["L"]
    n = len(r["added"])
    assert n == 51 and rb.idx == 51 and not rb.full
    # what add() stored is what the loop handed it, bit for bit (utils.py:120-128)
    obs_ring, next_ring = rb.stacks(0, n, 0), rb.stacks(0, n, 1)
    for i, (o, a, rew, no, nd) in enumerate(r["added"]):
        assert np.array_equal(obs_ring[i], o) and np.array_equal(next_ring[i], no), i
    assert np.array_equal(rb.actions[:n].cpu().numpy(), np.stack([t[1] for t in r["added"]]))
    assert np.array_equal(rb.rewards[:n, 0].cpu().numpy(), np.array([t[2] for t in r["added"]], np.float32))
    assert np.array_equal(rb.not_dones[:n, 0].cpu().numpy(), np.array([t[4] for t in r["added"]], np.float32))
    # the logger saw the reference's keys, on log steps only, and finite numbers
    assert AGENT_LOG_KEYS <= set(L.values)
    for key in AGENT_LOG_KEYS:
        steps = [s for s, _ in L.values[key]]
        assert steps and all(s % LOOP["log_interval"] == 0 and s >= LOOP["init_steps"] for s in steps), (key, steps)
        vals = [float(v.item() if torch.is_tensor(v) else v) for _, v in L.values[key]]
        assert np.all(np.isfinite(vals)), (key, vals)
    # full updates happened on the steps past the straight-line start of an episode; cpc-only ones still log the loss
    assert len(L.values["train/curl_loss"]) > len(L.values["train_critic/loss"]) > 0
    # eval.py:165: a fresh agent loads the three files and acts identically
    utils.set_seed_everywhere(123)
    other = CurlSacAgent(obs_shape=r["obs_shape"], action_shape=(2,), device=dev, augmentor=r["augmentor"], **r["kwargs"])
    probe = r["augmentor"].evaluation_augmentation(r["added"][-1][3])
    assert not np.array_equal(other.select_action(probe), agent.select_action(probe))
    other.load(r["model_dir"], "random_crop", 50)
    # (the loop's last iteration saved at step 50 and then trained once more: the files hold the state at the save)
    saved = torch.load(os.path.join(r["model_dir"], "random_crop_critic_50.pt"))
    assert any(not torch.equal(v.cpu(), agent.critic.state_dict()[k].cpu()) for k, v in saved.items())
    for k, v in saved.items():
        assert torch.equal(v.cpu(), other.critic.state_dict()[k].cpu()), k
        assert torch.equal(v.cpu(), other.critic_target.state_dict()[k].cpu()), k   # curl_sac.py:464
    agent.load(r["model_dir"], "random_crop", 50)
    with utils.eval_mode(agent, other):
        assert np.array_equal(other.select_action(probe), agent.select_action(probe))
        noise = torch.randn(1, 2, device=dev)
        assert np.array_equal(other.sample_action(r["added"][-1][3], noise=noise),
                              agent.sample_action(r["added"][-1][3], noise=noise))
    assert utils.module_hash(agent.actor) == utils.module_hash(other.actor)
    # the buffer chunks load back into a new buffer (utils.py:204-216)
    rb2 = utils.ReplayBuffer(obs_shape=r["pre_aug_obs_shape"], action_shape=(2,), capacity=64,
                             batch_size=LOOP["batch_size"], device=dev, augmentor=r["augmentor"])
    rb2.load(r["buffer_dir"])
    assert rb2.idx == 50
    assert np.array_equal(rb2.stacks(0, 50, 0), obs_ring[:50]) and np.array_equal(rb2.stacks(0, 50, 1), next_ring[:50])
    assert torch.equal(rb2.actions[:50], rb.actions[:50]) and torch.equal(rb2.not_dones[:50], rb.not_dones[:50])
    # training moved the parameters and kept the encoders tied (curl_sac.py:290)
    utils.set_seed_everywhere(7)
    fresh = CurlSacAgent(obs_shape=r["obs_shape"], action_shape=(2,), device=dev, augmentor=r["augmentor"], **r["kwargs"])
    assert not torch.equal(fresh.critic.encoder.convs[1].weight, agent.critic.encoder.convs[1].weight)
    assert agent.actor.encoder.convs[1].weight is agent.critic.encoder.convs[1].weight
    assert not torch.equal(fresh.critic_target.Q1.trunk[0].weight, agent.critic_target.Q1.trunk[0].weight)
