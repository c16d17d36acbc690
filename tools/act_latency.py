#!/usr/bin/env python3
"""Latency of the acting path (sample_action / select_action, B=1) -- SURVEY.md 8f rank 1."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curla_amd
dev = torch.device("cuda")
aug = curla_amd.RandomCrop((84, 84), (76, 76))
agent = curla_amd.CurlSacAgent((9, 76, 76), (2,), dev, aug, hidden_dim=1024)
obs = np.random.randint(0, 256, (9, 84, 84), dtype=np.uint8)
for _ in range(20):
    agent.sample_action(obs)
t0 = time.perf_counter()
for _ in range(200):
    agent.sample_action(obs)
t1 = time.perf_counter()
for _ in range(200):
    agent.select_action(aug.evaluation_augmentation(obs))
t2 = time.perf_counter()
print(f"sample_action {(t1 - t0) / 200 * 1e6:.0f} us   select_action {(t2 - t1) / 200 * 1e6:.0f} us (host->device->host, B=1)")

# replay ring insertion (utils.py:120-128): two uint8 frames + scalars per transition
rb = curla_amd.ReplayBuffer((9, 84, 84), (2,), 4096, 512, dev, aug)
nxt = np.random.randint(0, 256, (9, 84, 84), dtype=np.uint8)
for _ in range(20):
    rb.add(obs, [0.1, 0.2], 1.0, nxt, False)
t0 = time.perf_counter()
for _ in range(500):
    rb.add(obs, [0.1, 0.2], 1.0, nxt, False)
t1 = time.perf_counter()
print(f"ReplayBuffer.add {(t1 - t0) / 500 * 1e6:.0f} us per transition (2 x 63.5 KB frames + scalars; host enqueue, the GPU work is asynchronous)")
