"""Data-parallel gradient exchange, world_size 2 on CPU/gloo:
the flat gradient buckets all-reduced before each optimizer step must equal the
mean over ranks of the single-process reference gradients (golden minibatches
tiny.npz / tiny_rank1.npz), bucket by bucket."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests._util import load, sub

HP = dict(discount=0.99, init_temperature=0.1, alpha_lr=1e-4, alpha_beta=0.5, critic_tau=0.01, encoder_tau=0.05,
          log_interval=10 ** 9)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _set_grads(agent, fixture, phase):
    """Write one rank's reference gradients into the agent's flat gradient buffers."""
    with torch.no_grad():
        for name, g in sub(fixture, f"{phase}/grad/").items():
            if name == "W":
                agent.CURL.W.grad.copy_(g)
                continue
            if name == "log_alpha":
                agent.log_alpha.grad.copy_(g)
                continue
            mod = agent.actor if phase == "actor" else agent.critic
            p = dict(mod.named_parameters())[name]
            if name.endswith("encoder.fc.weight"):
                g = mod.encoder.fc.from_reference_layout(g)
            p.grad.copy_(g)


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import curla_amd
        from curla_amd import _lib
        curla_amd.set_seed_everywhere(1)
        aug = curla_amd.RandomCrop((34, 40), (28, 34))
        agent = curla_amd.CurlSacAgent((9, 28, 34), (2,), "cpu", aug, hidden_dim=64, **HP)
        agent.enable_data_parallel()
        fx = [load("tiny.npz"), load("tiny_rank1.npz")]
        mine = fx[rank]
        lay = agent._lay
        out = {}
        # critic bucket = [encoder | Q1 | Q2]
        _set_grads(agent, mine, "critic")
        agent._allreduce(agent._critic_gflat[lay["enc"][0]:lay["total"]])
        for name in sub(mine, "critic/grad/"):
            p = dict(agent.critic.named_parameters())[name]
            g = p.grad
            if name.endswith("encoder.fc.weight"):
                g = agent.critic.encoder.fc.to_reference_layout(g)
            out["critic/" + name] = g.clone().numpy()
        # actor bucket + the float64 log_alpha gradient
        _set_grads(agent, mine, "actor")
        _set_grads(agent, mine, "alpha")
        agent._allreduce(agent._actor_gflat, agent.log_alpha.grad)
        for name in sub(mine, "actor/grad/"):
            p = dict(agent.actor.named_parameters())[name]
            g = p.grad
            if name.endswith("encoder.fc.weight"):
                g = agent.actor.encoder.fc.to_reference_layout(g)
            out["actor/" + name] = g.clone().numpy()
        out["alpha/log_alpha"] = agent.log_alpha.grad.clone().numpy()
        # cpc bucket = [W | encoder]
        _set_grads(agent, mine, "cpc")
        agent._allreduce(agent._critic_gflat[0:lay["enc"][1]])
        for name in sub(mine, "cpc/grad/"):
            if name == "W":
                out["cpc/W"] = agent.CURL.W.grad.clone().numpy()
                continue
            p = dict(agent.critic.named_parameters())[name]
            g = p.grad
            if name.endswith("encoder.fc.weight"):
                g = agent.critic.encoder.fc.to_reference_layout(g)
            out["cpc/" + name] = g.clone().numpy()
        # launch/all-reduce schedule of a whole update (nothing is computed under the trace hook)
        n_calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, **k: (n_calls.append(t.numel()), real(t, **k))[1]
        _lib.set_trace_hook(lambda n, a: None)
        rb = curla_amd.ReplayBuffer((9, 34, 40), (2,), 16, 8, "cpu", aug)
        for _ in range(10):
            rb.add(np.zeros((9, 34, 40), np.uint8), [0, 0], 0.0, np.zeros((9, 34, 40), np.uint8), False)

        class L:
            def log(self, *a, **k):
                pass
        np.random.seed(rank)
        agent.update(rb, L(), 0)
        agent.update(rb, L(), 1)
        _lib.set_trace_hook(None)
        dist.all_reduce = real
        q.put((rank, out, n_calls, dict(lay)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, "ERR: " + traceback.format_exc(), None, None))
        raise e


@pytest.mark.timeout(300)
def test_two_rank_gradient_mean_matches_reference():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
    g0, g1 = load("tiny.npz"), load("tiny_rank1.npz")
    for rank, out, n_calls, lay in results:
        for key, got in out.items():
            phase, name = key.split("/", 1)
            want = 0.5 * (g0[f"{phase}/grad/{name}"].astype(np.float64) + g1[f"{phase}/grad/{name}"].astype(np.float64))
            err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
            assert err < 1e-6, (rank, key, err)
        # even step: critic bucket, actor bucket + log_alpha, cpc bucket; odd step: critic + cpc
        enc_q = lay["total"] - lay["enc"][0]
        w_enc = lay["enc"][1]
        actor_n = n_calls[1]
        assert n_calls == [enc_q, actor_n, 1, w_enc, enc_q, w_enc], n_calls
    # both ranks end with identical reduced gradients
    for k in results[0][1]:
        assert np.array_equal(results[0][1][k], results[1][1][k]), k
