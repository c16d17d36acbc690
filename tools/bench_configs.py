#!/usr/bin/env python3
"""update() throughput on the other BASELINE.json configurations (evidence for DESIGN.md; bench.py stays
on configs[1]):  c3 = pixel_sac (identity aug, 84x84x9, no CURL), c5 = 168x168x12, 6 layers, color_jiggle,
B=1024.  Prints one JSON line per configuration."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curla_amd  # noqa: E402

PEAK = 157.3e12


def conv_flops(c, hw, layers):
    h, w = (hw[0] - 3) // 2 + 1, (hw[1] - 3) // 2 + 1
    out = [2.0 * h * w * 32 * c * 9]
    for _ in range(layers - 1):
        h, w = h - 2, w - 2
        out.append(2.0 * h * w * 32 * 32 * 9)
    return out


class L:
    def log(self, *a, **k):
        pass


def run(name, obs_shape, aug_name, layers, B, pixel_sac, cap, steps, warm):
    dev = torch.device("cuda")
    curla_amd.set_seed_everywhere(1)
    aug = curla_amd.make_augmentor(aug_name, obs_shape[1:])
    agent = curla_amd.CurlSacAgent(obs_shape, (2,), dev, aug, hidden_dim=1024, init_temperature=0.1, alpha_lr=1e-4,
                                   alpha_beta=0.5, critic_tau=0.01, encoder_tau=0.05, num_layers=layers,
                                   pixel_sac=pixel_sac, log_interval=10 ** 9)
    rb = curla_amd.ReplayBuffer(obs_shape, (2,), cap, B, dev, aug)
    rb._obs_store.random_(0, 256)
    rb._next_store.random_(0, 256)
    rb.actions.uniform_(-1, 1)
    rb.rewards.normal_()
    rb.not_dones.fill_(1.0)
    rb.idx, rb.full = 0, True
    step = 0
    for _ in range(warm):
        agent.update(rb, L(), step)
        step += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.update(rb, L(), step)
        step += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    f = conv_flops(obs_shape[0], obs_shape[1:], layers)
    fc, f1 = sum(f), f[0]
    if pixel_sac:  # even steps 4 fwd, odd 3; 1 backward
        per = B * (3.5 * fc + (2 * fc - f1))
    else:
        per = B * (5 * fc + 2 * (2 * fc - f1))
    print(json.dumps({"config": name, "batch": B, "ms_per_update": 1e3 * dt, "updates_per_s": 1 / dt,
                      "transitions_per_s": B / dt, "conv_gflop_per_update": per / 1e9,
                      "conv_roofline_frac": per / dt / PEAK}), flush=True)
    del agent, rb
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run("c3 pixel_sac 84x84x9 identity B=512", (9, 84, 84), "identity", 4, 512, True, 20000, 100, 10)
    run("c5 168x168x12 L=6 color_jiggle B=1024", (12, 168, 168), "color_jiggle", 6, 1024, False, 4096, 10, 2)
    run("c5 geometry, identity aug B=1024", (12, 168, 168), "identity", 6, 1024, False, 4096, 10, 2)
