#!/usr/bin/env python3
"""Host-side enqueue time of update() vs GPU time (is the Python host ahead of the GPU?).
python tools/host_overhead.py [--graphs]   (--graphs: CurlSacAgent.enable_update_graphs, updates replayed from hipGraphs)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curla_amd
dev = torch.device("cuda")
curla_amd.set_seed_everywhere(1)
aug = curla_amd.RandomCrop((84, 84), (76, 76))
agent = curla_amd.CurlSacAgent((9, 76, 76), (2,), dev, aug, hidden_dim=1024, log_interval=10 ** 9)
rb = curla_amd.ReplayBuffer((9, 84, 84), (2,), 20000, 512, dev, aug)
rb._obs_store.random_(0, 256); rb._next_store.random_(0, 256)
rb.actions.uniform_(-1, 1); rb.rewards.normal_(); rb.not_dones.fill_(1.0); rb.idx, rb.full = 0, True
class L:
    def log(self, *a, **k): pass
if "--graphs" in sys.argv:
    agent.enable_update_graphs(rb)
step = 1  # (never a logging step: log_interval is 1e9 and step 0 is skipped)
for _ in range(20):
    agent.update(rb, L(), step); step += 1
torch.cuda.synchronize()
for n in (3, 3, 3, 3):  # (short bursts: with 2 graphs per kind the host may run 3 updates ahead without waiting)
    t0, c0 = time.perf_counter(), time.process_time()
    for _ in range(n):
        agent.update(rb, L(), step); step += 1
    t1, c1 = time.perf_counter(), time.process_time()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} updates: host enqueue {(t1 - t0) / n * 1e3:.3f} ms/update wall, {(c1 - c0) / n * 1e3:.3f} ms/update CPU; "
          f"total {(t2 - t0) / n * 1e3:.3f} ms/update", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    agent.update(rb, L(), step); step += 1
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
