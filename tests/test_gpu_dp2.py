"""Data parallelism with TWO LIVE RANKS on the one MI355X a test box has (BASELINE configs[3]).

RCCL refuses two ranks on one device, so the two ranks -- fresh ``spawn`` processes, each with its own HIP context on
``cuda:0``, its own replay-ring shard, its own sampling / policy-noise streams and DIFFERENT initial parameters --
exchange through a gloo group (the agent stages the buckets through the host for a backend without a device path:
``CurlSacAgent._dp_staged``).  Everything else is the N-GPU code path: ``enable_data_parallel`` (rank-0 broadcast,
replica checksum), the real kernels producing the gradients, three buckets per update, one ``optimizer.step()`` per
bucket on the averaged gradients.

What is asserted:
 (i)   after 6 updates (even and odd steps, blocking AND overlapped schedule) the two ranks' parameters, targets, Adam
       moments and log_alpha are bit-identical, ``check_replicas()`` stayed silent on every step, and the result is NOT
       what rank 0 computes alone (the exchange did something);
 (ii)  at step 0 the reduced buckets equal, to 1e-6, the mean of the two ranks' SINGLE-PROCESS gradients on the same
       minibatches (a third, non-distributed agent in the parent process fed rank r's ring, indices and noise), and, to
       1e-4, the mean of the oracle's gradients for the two minibatches (the parity definition of SURVEY.md 8e);
 (iii) the encoder gradients are reduced ONCE per update (inside the [W | encoder] bucket) and consumed by both
       ``encoder_optimizer`` and ``cpc_optimizer``.
"""
import os
import socket
import traceback

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

IN_HW, OUT_HW, B, HIDDEN, LAYERS = (40, 44), (32, 36), 16, 96, 4
HP = dict(discount=0.99, init_temperature=0.1, alpha_lr=1e-4, alpha_beta=0.5, actor_lr=1e-3, actor_beta=0.9,
          actor_log_std_min=-10, actor_log_std_max=2, actor_update_freq=2, critic_lr=1e-3, critic_beta=0.9,
          critic_tau=0.01, critic_target_update_freq=2, encoder_feature_dim=50, encoder_lr=1e-3, encoder_tau=0.05,
          num_layers=LAYERS, num_filters=32, log_interval=1)
# the step-0 probe: learning rates and taus zero, so every phase of every evaluation sees the broadcast parameters
HP0 = dict(HP, alpha_lr=0.0, actor_lr=0.0, critic_lr=0.0, encoder_lr=0.0, critic_tau=0.0, encoder_tau=0.0)
N_FILL, CAPACITY = 24, 64


class _Log:
    def __init__(self):
        self.s = {}

    def log(self, k, v, step, n=1):
        self.s[k] = float(v.item() if isinstance(v, torch.Tensor) else v)

    def log_histogram(self, *a, **k):
        pass

    log_param = log_image = log_histogram


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_data(rank):
    """Rank r's transitions (what its environment workers would have added to its shard)."""
    rs = np.random.RandomState(100 + rank)
    return dict(obs=rs.randint(0, 256, (N_FILL, 9) + IN_HW, dtype=np.uint8),
                nxt=rs.randint(0, 256, (N_FILL, 9) + IN_HW, dtype=np.uint8),
                act=rs.uniform(-1, 1, (N_FILL, 2)).astype(np.float32), rew=rs.randn(N_FILL).astype(np.float32),
                done=(np.arange(N_FILL) % 7) == 6)


def _make(hp, seed, world=1):
    import curla_amd
    dev = torch.device("cuda", 0)
    curla_amd.set_seed_everywhere(seed)
    aug = curla_amd.RandomCrop(IN_HW, OUT_HW)
    agent = curla_amd.CurlSacAgent((9,) + OUT_HW, (2,), dev, aug, hidden_dim=HIDDEN, **hp)
    rb = curla_amd.ReplayBuffer((9,) + IN_HW, (2,), CAPACITY // world, B, dev, aug)  # the rank's shard of the ring
    return agent, rb, aug


def _fill(rb, d):
    rb.add_batch(d["obs"], d["act"], d["rew"], d["nxt"], d["done"])


def _named_grads(agent):
    """Every gradient an optimizer consumes, by reference name and in the reference's tensor layouts."""
    out = {}
    for tag, mod in (("critic", agent.critic), ("actor", agent.actor)):
        for n, p in mod.named_parameters():
            if tag == "actor" and ".convs." in n:
                continue
            g = p.grad
            if n.endswith("encoder.fc.weight"):
                g = mod.encoder.fc.to_reference_layout(g)
            out[f"{tag}/{n}"] = g.detach().cpu().clone()
    out["W"] = agent.CURL.W.grad.detach().cpu().clone()
    out["log_alpha"] = agent.log_alpha.grad.detach().cpu().clone()
    return out


def _state(agent):
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}  # noqa: E731
    return dict(actor=cpu(agent.actor.state_dict()), critic=cpu(agent.critic.state_dict()),
                target=cpu(agent.critic_target.state_dict()), W=agent.CURL.W.detach().cpu().clone(),
                log_alpha=agent.log_alpha.detach().cpu().clone())


def _replica_bits(agent):
    """Everything that must be identical on all ranks, as host tensors."""
    out = dict(critic=agent._critic_flat, target=agent._target_flat, actor=agent._actor_flat,
               log_alpha=agent.log_alpha.detach())
    for name, opt in (("critic", agent.critic_optimizer), ("actor", agent.actor_optimizer),
                      ("encoder", agent.encoder_optimizer), ("cpc", agent.cpc_optimizer)):
        out[f"m/{name}"], out[f"v/{name}"] = opt._m, opt._v
        out[f"steps/{name}"] = torch.tensor(opt._steps)
    st = agent.log_alpha_optimizer.state[agent.log_alpha]
    out["m/log_alpha"], out["v/log_alpha"] = st["exp_avg"], st["exp_avg_sq"]
    return {k: v.detach().cpu().clone() for k, v in out.items()}


def _probe(rank, world):
    """One even update with lr = 0 through the public ``update()``: what went into each bucket on this rank, what came
    out of the exchange, and everything the parent needs to re-evaluate this rank's minibatch elsewhere."""
    import curla_amd
    agent, rb, _ = _make(HP0, 1 + rank, world)     # ranks start from DIFFERENT parameters ...
    before = agent._replica_checksum().cpu()
    agent.enable_data_parallel(overlap=False, check_every=1)   # ... and the rank-0 broadcast makes them replicas
    after = agent._replica_checksum().cpu()
    _fill(rb, _shard_data(rank))
    curla_amd.set_seed_everywhere(11 + rank)       # this rank's sampling and policy-noise streams
    rec = dict(calls=[], draws=[])
    real_draw, real_ar = rb.draw_indices, agent._allreduce

    def draw():
        out = real_draw()
        rec["draws"].append(out)
        return out

    def allreduce(*buckets, **kw):
        ws = agent._ws(B)
        local = _named_grads(agent)
        real_ar(*buckets, **kw)
        rec["calls"].append(dict(
            sizes=[int(t.numel()) for t in buckets], local=local, reduced=_named_grads(agent),
            noise=ws.noise.detach().cpu().clone(),
            branches=[(a[:B].permute(0, 3, 1, 2) > 0).cpu() for a in ws.acts_main]))
    rb.draw_indices, agent._allreduce = draw, allreduce
    agent.update(rb, _Log(), 0)
    torch.cuda.synchronize()
    assert len(rec["draws"]) == 1 and len(rec["calls"]) == 3, (len(rec["draws"]), len(rec["calls"]))
    return dict(before=before, after=after, state=_state(agent), lay=dict(agent._lay), indices=rec["draws"][0],
                phases=dict(zip(("critic", "actor", "cpc"), rec["calls"])))


def _train(rank, world, overlap, steps=6):
    """``steps`` real updates through ``update()``; returns the replicated state bit for bit."""
    import curla_amd
    import torch.distributed as dist
    agent, rb, _ = _make(HP, 1 + rank, world)
    agent.enable_data_parallel(overlap=overlap, check_every=1)   # check_replicas() on EVERY update: raises on drift
    _fill(rb, _shard_data(rank))
    curla_amd.set_seed_everywhere(11 + rank)
    sizes = []
    real = dist.all_reduce

    def spy(t, *a, **k):
        sizes.append(int(t.numel()))
        return real(t, *a, **k)
    dist.all_reduce = spy
    try:
        L = _Log()
        for step in range(steps):
            agent.update(rb, L, step)
        torch.cuda.synchronize()
        agent.check_replicas()
    finally:
        dist.all_reduce = real
    assert not agent._dp_pending
    return dict(bits=_replica_bits(agent), sizes=sizes, lay=dict(agent._lay), losses=dict(L.s),
                enc_cut=agent._grad_offset(agent.critic.encoder.fc.weight, agent._critic_gflat),
                actor_bucket=int(agent._actor_gbucket.numel()))


def _solo(steps=6):
    """Rank 0's run WITHOUT the exchange (same seeds, same shard)."""
    import curla_amd
    agent, rb, _ = _make(HP, 1, 2)
    _fill(rb, _shard_data(0))
    curla_amd.set_seed_everywhere(11)
    L = _Log()
    for step in range(steps):
        agent.update(rb, L, step)
    torch.cuda.synchronize()
    return _replica_bits(agent)


def _worker(rank, world, port, out_dir):
    """Entry point of a rank (a fresh ``spawn`` process: nothing of the parent's HIP state is inherited)."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        out = dict(probe=_probe(rank, world), blocking=_train(rank, world, False), overlapped=_train(rank, world, True))
        if rank == 0:
            out["solo"] = _solo()
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        with open(os.path.join(out_dir, f"rank{rank}.err"), "w") as f:
            f.write(traceback.format_exc())
        raise


@pytest.fixture(scope="module")
def two_ranks(tmp_path_factory):
    import torch.multiprocessing as mp
    out_dir = str(tmp_path_factory.mktemp("dp2"))
    ctx = mp.get_context("spawn")
    port, world = _free_port(), 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, out_dir)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    errs = []
    for r, p in enumerate(procs):
        if p.is_alive():
            p.kill()
            errs.append(f"rank {r}: still running after 600 s")
        path = os.path.join(out_dir, f"rank{r}.err")
        if os.path.exists(path):
            errs.append(f"rank {r}:\n" + open(path).read())
        elif p.exitcode != 0:
            errs.append(f"rank {r}: exit code {p.exitcode}")
    assert not errs, "\n".join(errs)
    return [torch.load(os.path.join(out_dir, f"rank{r}.pt"), weights_only=False) for r in range(world)]


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def test_rank0_broadcast_makes_replicas(two_ranks):
    p0, p1 = two_ranks[0]["probe"], two_ranks[1]["probe"]
    assert not torch.equal(p0["before"], p1["before"]), "the ranks were meant to start from different parameters"
    assert torch.equal(p0["before"], p0["after"]), "rank 0 keeps its parameters"
    assert torch.equal(p1["after"], p0["after"]), "rank 1 takes rank 0's"
    for k in ("actor", "critic", "target"):
        for n in p0["state"][k]:
            assert torch.equal(p0["state"][k][n], p1["state"][k][n]), (k, n)


@pytest.mark.parametrize("schedule", ["blocking", "overlapped"])
def test_two_live_ranks_stay_bitwise_replicas(two_ranks, schedule):
    r0, r1 = two_ranks[0][schedule], two_ranks[1][schedule]
    assert r0["bits"].keys() == r1["bits"].keys()
    for k in r0["bits"]:
        assert torch.equal(r0["bits"][k], r1["bits"][k]), f"{schedule}: ranks differ in {k} after 6 updates"
    assert bool(torch.isfinite(r0["bits"]["critic"]).all()) and float(r0["bits"]["critic"].abs().sum()) > 0
    # ... the exchange changed the trajectory (rank 0 alone ends elsewhere), and both schedules reduce the same elements
    solo = two_ranks[0]["solo"]
    assert not torch.equal(solo["critic"], r0["bits"]["critic"]) and not torch.equal(solo["actor"], r0["bits"]["actor"])
    other = two_ranks[0]["overlapped" if schedule == "blocking" else "blocking"]["bits"]
    for k in r0["bits"]:
        assert torch.equal(r0["bits"][k], other[k]), f"blocking and overlapped schedules differ in {k}"
    # the ranks logged their OWN minibatch's losses: different data, so different numbers
    assert two_ranks[0][schedule]["losses"] != two_ranks[1][schedule]["losses"]


@pytest.mark.parametrize("schedule", ["blocking", "overlapped"])
def test_encoder_gradients_are_reduced_once_and_consumed_twice(two_ranks, schedule):
    r = two_ranks[0][schedule]
    lay, sizes = r["lay"], r["sizes"]
    e0, e1, total = lay["enc"][0], lay["enc"][1], lay["total"]
    actor_n = int(r["bits"]["actor"].numel())
    assert r["actor_bucket"] == actor_n + 8  # [fc, ln | trunk | the 8 words log_alpha's float64 gradient rides in]
    # 6 updates: critic + cpc every step, the actor bucket on the 3 even steps, one 8-element replica check per update
    # and the final one.  Round 6: THREE gradient collectives per even update in the blocking schedule (log_alpha's
    # gradient had an 8-byte all-reduce of its own), FIVE in the overlapped one (was seven)
    if schedule == "blocking":
        per_even = [8, total - e0, actor_n + 8, e1]
        per_odd = [8, total - e0, e1]
    else:  # critic and cpc buckets in two pieces, [dense | convs]; the actor bucket whole, under the CURL phase
        ec = r["enc_cut"]
        per_even = [8, total - ec, ec - e0, actor_n + 8, e1 - ec, ec]
        per_odd = [8, total - ec, ec - e0, e1 - ec, ec]
    assert sizes == (per_even + per_odd) * 3 + [8], (schedule, sizes)
    # one exchange of the encoder's gradients per update -- and both optimizers that own the encoder stepped on it
    steps = r["bits"]
    assert steps["steps/encoder"].tolist() == [6] * len(steps["steps/encoder"])
    assert steps["steps/cpc"].tolist() == [6] * len(steps["steps/cpc"])
    assert set(steps["steps/critic"].tolist()) == {6} and set(steps["steps/actor"].tolist()) == {3}


def test_reduced_buckets_are_the_mean_of_single_process_and_oracle_gradients(two_ranks):
    import curla_amd
    from oracle import curla_oracle as O
    probes = [two_ranks[r]["probe"] for r in range(2)]
    phases = ("critic", "actor", "cpc")
    live = {"critic": lambda k: k.startswith("critic/"),
            "actor": lambda k: k.startswith("actor/") or k == "log_alpha",
            "cpc": lambda k: k == "W" or k.startswith("critic/encoder.")}
    report = []
    # -- the exchange itself: both ranks hold the same reduced gradients, = (g0 + g1) / 2 of what went in
    for ph in phases:
        c0, c1 = probes[0]["phases"][ph], probes[1]["phases"][ph]
        for k in filter(live[ph], c0["reduced"]):
            assert torch.equal(c0["reduced"][k], c1["reduced"][k]), (ph, k)
            mean = (c0["local"][k].double() + c1["local"][k].double()) / 2
            assert _rel(c0["reduced"][k], mean) <= 1e-6, (ph, k)
            assert not torch.equal(c0["local"][k], c1["local"][k]) or float(c0["local"][k].abs().max()) == 0, (ph, k)
    lay = probes[0]["lay"]
    assert probes[0]["phases"]["critic"]["sizes"] == [lay["total"] - lay["enc"][0]]
    assert len(probes[0]["phases"]["actor"]["sizes"]) == 1   # ONE exchange: log_alpha's gradient rides in the bucket
    # ... and comes out as today's float64 all-reduce would leave it, bit for bit: (a + b) / 2 in double
    la = [probes[r]["phases"]["actor"]["local"]["log_alpha"].double() for r in range(2)]
    assert torch.equal(probes[0]["phases"]["actor"]["reduced"]["log_alpha"].double(), (la[0] + la[1]) / 2)
    assert not torch.equal(la[0], la[1])
    assert probes[0]["phases"]["cpc"]["sizes"] == [lay["enc"][1]]   # [W | encoder] in ONE exchange

    # -- single-process gradients: a non-distributed agent in THIS process on rank r's ring, indices and noise
    st = probes[0]["state"]
    agent, _, aug = _make(HP0, 5)
    agent.critic.load_state_dict(st["critic"])
    agent.actor.load_state_dict(st["actor"])
    agent.critic_target.load_state_dict(st["target"])
    with torch.no_grad():
        agent.CURL.W.copy_(st["W"])
        agent.log_alpha.copy_(st["log_alpha"])
    single, oracle = [], []
    kw = dict(num_layers=LAYERS, log_std_min=-10, log_std_max=2)
    L = _Log()
    for r in range(2):
        d = _shard_data(r)
        rb = curla_amd.ReplayBuffer((9,) + IN_HW, (2,), CAPACITY // 2, B, torch.device("cuda", 0), aug)
        _fill(rb, d)
        idxs, offs = probes[r]["indices"]
        ph = probes[r]["phases"]
        nc, na = ph["critic"]["noise"], ph["actor"]["noise"]
        obs, act, rew, nxt, nd, ckw = rb.sample_cpc_refs(indices=(idxs, offs))
        got = {}
        agent.update_critic(obs, act, rew, nxt, nd, L, 0, noise=nc.cuda())
        got.update({k: v for k, v in _named_grads(agent).items() if live["critic"](k)})
        agent.update_actor_and_alpha(obs, L, 0, noise=na.cuda())
        got.update({k: v for k, v in _named_grads(agent).items() if live["actor"](k)})
        agent.update_cpc(ckw["obs_anchor"], ckw["obs_pos"], ckw, L, 0)
        cpc = {"cpc:" + k: v for k, v in _named_grads(agent).items() if live["cpc"](k)}
        got.update(cpc)
        single.append(got)
        for p_, tag in (("critic", ""), ("actor", ""), ("cpc", "cpc:")):
            for k in filter(live[p_], ph[p_]["local"]):
                e = _rel(ph[p_]["local"][k], got[tag + k])
                report.append((f"rank{r} {p_} local vs single-process {k}", e))
                assert e <= 1e-6, (r, p_, k, e)
        # -- the oracle on the same parameters, minibatch and noise (conv gradients along the device's ReLU branches)
        crop = lambda src, j: torch.from_numpy(O.random_crop(src[idxs], offs[2 * j], offs[2 * j + 1], OUT_HW)).float()  # noqa: E731
        o_obs, o_nxt, o_pos = crop(d["obs"], 0), crop(d["nxt"], 1), crop(d["obs"], 2)
        o_act, o_rew = torch.from_numpy(d["act"][idxs]), torch.from_numpy(d["rew"][idxs])[:, None]
        o_nd = torch.from_numpy(1.0 - d["done"][idxs].astype(np.float32))[:, None]
        ref = {}
        c = O.critic_phase(st["actor"], st["critic"], st["target"], st["log_alpha"], o_obs, o_act, o_rew, o_nxt, o_nd, nc,
                           discount=0.99, relu_branches=ph["critic"]["branches"], **kw)
        ref.update({"critic/" + k: v for k, v in c["grads"].items()})
        a = O.actor_phase(st["actor"], st["critic"], st["log_alpha"], o_obs, na, target_entropy=-2.0, **kw)
        ref.update({"actor/" + k: v for k, v in a["grads"].items()})
        ref["log_alpha"] = a["log_alpha_grad"]
        p = O.cpc_phase(st["critic"], st["target"], st["W"], o_obs, o_pos, num_layers=LAYERS,
                        relu_branches=ph["cpc"]["branches"])
        ref.update({"cpc:critic/" + k: v for k, v in p["grads"].items()})
        ref["cpc:W"] = p["W_grad"]
        oracle.append(ref)
    assert len(oracle[0]) == 24 + 10 + 1 + 12 + 1
    # -- reduced == mean over ranks of the single-process gradients (1e-6) == mean of the oracle's (1e-4)
    for p_, tag in (("critic", ""), ("actor", ""), ("cpc", "cpc:")):
        red = probes[0]["phases"][p_]["reduced"]
        for k in filter(live[p_], red):
            m_single = (single[0][tag + k].double() + single[1][tag + k].double()) / 2
            e = _rel(red[k], m_single)
            report.append((f"{p_} reduced vs mean of single-process {k}", e))
            assert e <= 1e-6, (p_, k, e)
            m_oracle = (torch.as_tensor(oracle[0][tag + k]).double() + torch.as_tensor(oracle[1][tag + k]).double()) / 2
            e = _rel(red[k], m_oracle.reshape(red[k].shape))
            report.append((f"{p_} reduced vs mean of oracle {k}", e))
            assert e <= 1e-4, (p_, k, e)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/dp2_parity.txt", "w") as f:
        for n, e in report:
            f.write(f"{n:90s} {e:.3e}\n")
