"""Data parallelism on the MI355X through RCCL (torch.distributed backend "nccl"), with the one GPU a test box
has: a world of ONE rank issuing the exact collectives an N-GPU run issues ("N=1 must equal the
single-GPU path bit-for-bit").  Covers ncclAvg on the float32 buckets, SUM + divide on the float64 log_alpha
gradient, the rank-0 broadcast, the replica checksum and both all-reduce schedules (overlapped / blocking)."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HP = dict(discount=0.99, init_temperature=0.1, alpha_lr=1e-4, alpha_beta=0.5, critic_tau=0.01, encoder_tau=0.05,
          log_interval=1)


class Log:
    def __init__(self):
        self.s = {}

    def log(self, k, v, step, n=1):
        self.s[k] = float(v.item() if isinstance(v, torch.Tensor) else v)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _run(mode, steps=4, graphs=False, B=16, hp=None):
    """mode None: plain single-GPU agent; else dict of enable_data_parallel kwargs."""
    import curla_amd
    dev = torch.device("cuda", 0)
    curla_amd.set_seed_everywhere(7)
    in_hw, out_hw = (40, 44), (32, 36)
    aug = curla_amd.RandomCrop(in_hw, out_hw)
    agent = curla_amd.CurlSacAgent((9,) + out_hw, (2,), dev, aug, hidden_dim=96, **(hp or HP))
    if mode is not None:
        agent.enable_data_parallel(single_rank_collectives=True, **mode)
    rb = curla_amd.ReplayBuffer((9,) + in_hw, (2,), 64, B, dev, aug)
    rs = np.random.RandomState(0)
    n = 40
    rb.add_batch(rs.randint(0, 256, (n, 9) + in_hw, dtype=np.uint8), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                 rs.randn(n).astype(np.float32), rs.randint(0, 256, (n, 9) + in_hw, dtype=np.uint8),
                 (np.arange(n) % 9) == 8)
    curla_amd.set_seed_everywhere(11)  # sampling and policy-noise streams
    if graphs:
        agent.enable_update_graphs(rb)
    L = Log()
    for step in range(steps):
        agent.update(rb, L, step)
    torch.cuda.synchronize()
    assert not agent._dp_pending
    _run.last_agent = agent
    return (agent._critic_flat.clone(), agent._target_flat.clone(), agent._actor_flat.clone(),
            agent.log_alpha.detach().clone(), dict(L.s))


def test_world1_rccl_update_is_bitwise_the_single_gpu_update(nccl_world1):
    dist = nccl_world1
    calls = []
    real = dist.all_reduce

    def spy(t, **k):
        calls.append((t.numel(), str(k.get("op")), bool(k.get("async_op"))))
        return real(t, **k)
    base = _run(None)
    dist.all_reduce = spy
    try:
        over = _run(dict(overlap=True, check_every=2))
        n_over = len(calls)
        block = _run(dict(overlap=False, check_every=0))
        n_block = len(calls) - n_over
    finally:
        dist.all_reduce = real
    for name, other in (("overlapped", over), ("blocking", block)):
        for what, a, b in zip(("critic", "target", "actor", "log_alpha"), base, other):
            assert torch.equal(a, b), f"{name} RCCL world-1 run differs from the single-GPU run in {what}"
        assert base[4] == other[4], name  # the logged losses too
    assert float(base[0].abs().sum()) > 0 and bool(torch.isfinite(base[0]).all())
    # 4 updates (2 even, 2 odd): overlapped = 2 pieces for the critic and cpc buckets, the actor bucket whole (log_alpha's
    # float64 gradient rides in it), + 2 replica checks; blocking = 1 per bucket: 5 and 3 gradient collectives per even update
    assert n_over == 2 * (2 + 1 + 2) + 2 * (2 + 2) + 2 and n_block == 2 * (1 + 1 + 1) + 2 * (1 + 1), (n_over, n_block)
    assert any("AVG" in op.upper() for _, op, _ in calls), "ncclAvg branch not exercised"
    assert any(a for _, _, a in calls[:n_over]) and not any(a for _, _, a in calls[n_over:])


def test_broadcast_and_drift_check_on_rccl(nccl_world1):
    import curla_amd
    dev = torch.device("cuda", 0)
    curla_amd.set_seed_everywhere(3)
    aug = curla_amd.RandomCrop((40, 44), (32, 36))
    agent = curla_amd.CurlSacAgent((9, 32, 36), (2,), dev, aug, hidden_dim=64, **HP)
    before = agent._replica_checksum().clone()
    agent.enable_data_parallel(single_rank_collectives=True)
    assert torch.equal(before, agent._replica_checksum())  # rank 0 broadcasts to itself: nothing moves
    agent.check_replicas()  # a world of one cannot diverge; the collective itself must run on RCCL


@pytest.mark.parametrize("overlap", [False, True])
def test_update_graphs_capture_the_collectives(nccl_world1, overlap):
    """Round 5: data-parallel updates over RCCL are captured into the update graphs WITH their all-reduces (the
    replica check stays on the host, in front of the replay).  A world-1 RCCL run with graphs enabled must be the eager
    data-parallel run -- and the plain single-GPU run -- bit for bit, and the replayed updates must make no
    ``all_reduce`` call from the host."""
    dist = nccl_world1
    hp = dict(HP, log_interval=7)
    base = _run(None, steps=16, hp=hp)
    eager = _run(dict(overlap=overlap, check_every=0), steps=16, hp=hp)
    calls = []
    real = dist.all_reduce

    def spy(t, **k):
        calls.append(t.numel())
        return real(t, **k)
    dist.all_reduce = spy
    try:
        graph = _run(dict(overlap=overlap, check_every=0), steps=16, graphs=True, hp=hp)
    finally:
        dist.all_reduce = real
    agent = _run.last_agent
    assert agent._graphs and all(g["graph"] is not None for r in agent._graphs.values() for g in r)
    for what, a, b, c in zip(("critic", "target", "actor", "log_alpha"), base, eager, graph):
        assert torch.equal(a, b), f"eager DP differs from single-GPU in {what}"
        assert torch.equal(a, c), f"graphed DP differs from single-GPU in {what}"
    assert base[4] == graph[4]
    # 16 updates: 0, 7, 14 log (eager), 1, 2 warm up, 3, 4, 5, 6 capture (their collectives are recorded: the host
    # still calls all_reduce while capturing), 8 .. 13 and 15 replay with no host-side collective call
    per_even, per_odd = ((2 + 1 + 2), (2 + 2)) if overlap else ((1 + 1 + 1), (1 + 1))
    host_side = [0, 1, 2, 3, 4, 5, 6, 7, 14]
    want = sum(per_even if s % 2 == 0 else per_odd for s in host_side)
    assert len(calls) == want, (len(calls), want)
