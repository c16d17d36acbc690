#!/usr/bin/env python3
"""How does update() time evolve under sustained load (DVFS)?  Prints ms/update per 50-step window."""
import os, sys, time
import numpy as np, torch
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
step = 0
for _ in range(10):
    agent.update(rb, L(), step); step += 1
torch.cuda.synchronize()
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    t0 = time.perf_counter()
    for _ in range(50):
        agent.update(rb, L(), step); step += 1
    torch.cuda.synchronize()
    print(f"window {w:2d}: {(time.perf_counter() - t0) * 20:.3f} ms/update", flush=True)
