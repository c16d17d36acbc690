"""``CurlSacAgent.enable_update_graphs``: whole updates replayed from captured hipGraphs ("graph-capturable";
8d: the metric's steady state) must be the eager updates, bit for bit.

Two agents built from the same seeds run the same 14 steps on the same ring contents -- one eagerly, one with update
graphs enabled (one warm-up update per kind, then capture, then replays; ``log_interval = 5`` puts eager logging steps
in between the replays) -- and must end with identical parameters, targets, Adam moments and step counts, log_alpha,
generator state and NumPy stream position.  During a replayed update the host makes NO kernel call through the C ABI."""
import collections

import numpy as np
import pytest
import torch

from tests.test_gpu_agent import HP, NullLogger

pytestmark = pytest.mark.gpu


def _run(graphs, steps=14, pixel_sac=False, detach_encoder=False, edit=None):
    """``edit`` = (step, fn): ``fn(agent)`` is applied before that step's update (a hyper-parameter edited mid-run)."""
    import curla_amd
    import curla_amd.ops as ops_mod
    import curla_amd.optim as optim_mod
    from curla_amd import _lib
    torch.manual_seed(5)
    np.random.seed(5)
    dev = torch.device("cuda")
    B, hidden = 256, 64
    C, in_hw, out_hw = 9, (40, 44), (32, 36)
    aug = curla_amd.RandomCrop(in_hw, out_hw)
    agent = curla_amd.CurlSacAgent((C,) + out_hw, (2,), dev, aug, hidden_dim=hidden, pixel_sac=pixel_sac,
                                   detach_encoder=detach_encoder, **{**HP, "log_interval": 5})
    rb = curla_amd.ReplayBuffer((C,) + in_hw, (2,), 512, B, dev, aug)
    rs = np.random.RandomState(6)
    n = 400
    rb.add_batch(rs.randint(0, 256, (n, C) + in_hw, dtype=np.uint8), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                 rs.randn(n).astype(np.float32), rs.randint(0, 256, (n, C) + in_hw, dtype=np.uint8),
                 (np.arange(n) % 7) == 6)
    if graphs:
        agent.enable_update_graphs(rb)
    agent._test_rb = rb
    calls_per_step = []
    real_call = _lib.call
    counter = collections.Counter()

    def traced(name, *a):
        counter[name] += 1
        return real_call(name, *a)
    mods = (ops_mod, optim_mod)
    for m in mods:
        m.call = traced
    L = NullLogger()
    try:
        for step in range(steps):
            counter.clear()
            if edit is not None and edit[0] == step:
                edit[1](agent)
            agent.update(rb, L, step)
            calls_per_step.append(sum(counter.values()))
        torch.cuda.synchronize()
    finally:
        for m in mods:
            m.call = real_call
    state = {"critic": agent._critic_flat, "target": agent._target_flat, "actor": agent._actor_flat,
             "log_alpha": agent.log_alpha.detach(), "rng": torch.cuda.get_rng_state(dev),
             "scalars": agent._ws(B).scalars}
    for name, opt in (("critic", agent.critic_optimizer), ("actor", agent.actor_optimizer),
                      ("encoder", agent.encoder_optimizer), ("cpc", agent.cpc_optimizer)):
        state[name + "_m"], state[name + "_v"] = opt._m, opt._v
        state[name + "_steps"] = torch.tensor(opt._steps)
        sd = opt.state_dict()["state"]
        state[name + "_sd_steps"] = torch.tensor([float(v["step"]) for v in sd.values()])
    la = agent.log_alpha_optimizer.state[agent.log_alpha]
    state["la_m"], state["la_v"], state["la_step"] = la["exp_avg"], la["exp_avg_sq"], la["step"]
    state = {k: v.detach().cpu().clone() for k, v in state.items()}
    return state, calls_per_step, float(np.random.rand()), dict(L.scalars), agent


@pytest.mark.parametrize("pixel_sac", [False, True])
def test_graph_replay_is_the_eager_update_bit_for_bit(pixel_sac):
    eager, calls_e, np_e, logs_e, _ = _run(False, pixel_sac=pixel_sac)
    graph, calls_g, np_g, logs_g, agent = _run(True, pixel_sac=pixel_sac)
    assert min(calls_e) > 15  # every eager update makes its launches from the host
    # steps 0, 5, 10 log (eager); 1, 2 warm the two kinds up; 3, 4, 6, 7 capture (two graphs per kind, used in rotation:
    # their launches are recorded into the graphs); everything after replays with no host-side kernel call at all
    replayed = [8, 9, 11, 12, 13]
    assert [calls_g[s] for s in replayed] == [0] * len(replayed), calls_g
    assert all(calls_g[s] > 15 for s in (0, 1, 2, 3, 4, 5, 6, 7, 10)), calls_g
    assert len(agent._graphs) == 2 and all(len(r) == 2 and all(g["graph"] is not None for g in r)
                                           for r in agent._graphs.values())
    assert np_e == np_g  # the NumPy stream (sample indices, crop offsets) is where the eager run left it
    assert logs_e == logs_g
    for k in eager:
        assert torch.equal(eager[k], graph[k]), k
    assert float(eager["critic_steps"][0]) == 14 and float(eager["actor_steps"][0]) == 7


def test_detach_encoder_graphs_keep_the_convs_step_counts_behind():
    """Under detach_encoder the critic's convs have no gradient at the critic's step (curl_sac.py:358): torch's Adam
    skips them, so in ``critic_optimizer`` their step counts stay at zero while fc / ln / Q count on.  The captured
    step must take its bias-correction factors from the LIVE parameters' count and the replay bookkeeping must not
    advance (or create state for) the skipped ones."""
    eager, _, np_e, logs_e, _ = _run(False, detach_encoder=True)
    graph, calls_g, np_g, logs_g, agent = _run(True, detach_encoder=True)
    assert [calls_g[s] for s in (8, 9, 11, 12, 13)] == [0] * 5, calls_g
    assert np_e == np_g and logs_e == logs_g
    for k in eager:
        assert torch.equal(eager[k], graph[k]), k
    steps = eager["critic_steps"].tolist()
    assert set(steps) == {0, 14} and steps.count(0) == 8  # 4 conv layers x (weight, bias) never stepped by the critic
    assert len(eager["critic_sd_steps"]) == len(steps) - 8  # ... and have no Adam state, as with torch.optim.Adam


def test_editing_a_captured_value_recaptures_instead_of_replaying_stale_arguments():
    """discount, the taus, betas / eps are kernel ARGUMENTS inside a captured graph.  An edit after capture must not be
    ignored: the graphs are dropped and captured again (eager updates in between), and the run equals the eager run
    with the same edit at the same step."""
    def edit(agent):
        agent.discount = 0.9
        agent.critic_tau = 0.02
        agent.critic_optimizer.param_groups[0]["betas"] = (0.8, 0.99)
        agent.actor_optimizer.param_groups[0]["lr"] = 3e-4   # (lr travels as data: needs no re-capture, must still act)
    eager, _, np_e, logs_e, _ = _run(False, steps=20, edit=(11, edit))
    graph, calls_g, np_g, logs_g, agent = _run(True, steps=20, edit=(11, edit))
    assert calls_g[9] == 0 and calls_g[11] > 15  # replaying before the edit, eager (warm-up of the new graphs) after
    assert calls_g[19] == 0                       # ... and replaying again at the end
    assert np_e == np_g and logs_e == logs_g
    for k in eager:
        assert torch.equal(eager[k], graph[k]), k
    plain, _, _, _, _ = _run(False, steps=20)
    assert not torch.equal(plain["critic"], eager["critic"])  # (the edit matters)


def test_load_checkpoint_drops_the_captured_graphs(tmp_path):
    """The captured log_alpha step holds the addresses of its Adam moments; ``load_state_dict`` replaces those
    tensors.  A resume must therefore re-capture -- and continue bit for bit like an eager agent resumed the same way."""
    out = []
    for graphs in (False, True):
        _, _, _, _, agent = _run(graphs, steps=10)
        path = str(tmp_path / ("g.pt" if graphs else "e.pt"))
        agent.save_checkpoint(path, 10)
        if graphs:  # steps 3, 4, 6, 7 captured
            assert sum(len(r) for r in agent._graphs.values()) == 4
        assert agent.load_checkpoint(path) == 10
        if graphs:
            assert agent._graphs == {}  # still enabled, nothing captured
        L = NullLogger()
        for step in range(10, 20):
            agent.update(agent._test_rb, L, step)
        torch.cuda.synchronize()
        if graphs:
            assert sum(len(r) for r in agent._graphs.values()) == 4
        la = agent.log_alpha_optimizer.state[agent.log_alpha]
        out.append({k: v.detach().cpu().clone() for k, v in dict(
            critic=agent._critic_flat, target=agent._target_flat, actor=agent._actor_flat,
            log_alpha=agent.log_alpha.detach(), la_m=la["exp_avg"], la_v=la["exp_avg_sq"]).items()})
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k


def test_unsupported_setups_are_refused():
    import curla_amd
    dev = torch.device("cuda")
    aug = curla_amd.ColorJiggle((20, 24))
    agent = curla_amd.CurlSacAgent((12, 20, 24), (2,), dev, aug, hidden_dim=32, **HP)
    rb = curla_amd.ReplayBuffer((12, 20, 24), (2,), 64, 8, dev, aug)
    with pytest.raises(ValueError):
        agent.enable_update_graphs(rb)  # float augmentation: parameters staged per call


def test_bench_graphs_flag_runs_and_reports():
    """``bench.py --graphs``: the timed updates are graph replays, the dominant kernel's HIP-event timing comes from
    eager updates after the timed region; one JSON line, nothing else on stdout."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "bench.py", "--config", "c3", "--steps", "6", "--warmup", "2", "--graphs",
                        "--no-cpu-baseline", "--capacity", "4096", "--clock-warmup-s", "0"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["config"]["update_graphs"] is True and d["steps"] == 6 and d["value"] > 0
    r = d["roofline"]
    assert r["launches"] > 0 and r["avg_launch_ms"] > 0 and "after the timed region" in r["launch_timing"]
    assert 0.3 < r["frac"] < 1.0
