#!/usr/bin/env python3
"""Training-like soak: environment steps (ReplayBuffer.add of uint8 stacks that share frames, as FrameStack produces
them) interleaved with updates, on the plain ring and on the de-duplicating store, with a state check every 100
updates.  Usage: tools/soak.py [n_updates] [--graphs]
--graphs: a third run on the plain ring with CurlSacAgent.enable_update_graphs (updates replayed from captured hipGraphs
while the environment keeps writing into the ring between them): its checksum must equal the eager runs' as well."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curla_amd  # noqa: E402


class L:
    def log(self, *a, **k):
        pass

    log_histogram = log_param = log_image = log


def run(dedup, n_updates, graphs=False):
    dev = torch.device("cuda")
    curla_amd.set_seed_everywhere(3)
    aug = curla_amd.RandomCrop((84, 84), (76, 76))
    agent = curla_amd.CurlSacAgent((9, 76, 76), (2,), dev, aug, hidden_dim=256, log_interval=50)
    rb = curla_amd.ReplayBuffer((9, 84, 84), (2,), 600, 64, dev, aug, dedup_frames=dedup)
    if graphs:
        agent.enable_update_graphs(rb)
    rs = np.random.RandomState(0)
    frames = [rs.randint(0, 256, (3, 84, 84), dtype=np.uint8) for _ in range(3)]
    step = 0
    for t in range(n_updates + 200):
        obs = np.concatenate(frames, 0)
        frames = frames[1:] + [rs.randint(0, 256, (3, 84, 84), dtype=np.uint8)]
        nxt = np.concatenate(frames, 0)
        done = (t % 97) == 96
        rb.add(obs, rs.uniform(-1, 1, 2).astype(np.float32), float(rs.randn()), nxt, done)
        if done:
            frames = [rs.randint(0, 256, (3, 84, 84), dtype=np.uint8)] * 3
        if t >= 200:
            agent.update(rb, L(), step)
            step += 1
            if step % 100 == 0:
                torch.cuda.synchronize()
                ok = all(bool(torch.isfinite(b).all()) for b in (agent._critic_flat, agent._target_flat, agent._actor_flat))
                print(f"dedup={dedup} graphs={graphs} update {step}: finite={ok} |critic|={float(agent._critic_flat.abs().mean()):.5f} "
                      f"alpha={float(agent.alpha):.5f}", flush=True)
                assert ok
    return agent._critic_flat.double().sum().item()


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    n = int(args[0]) if args else 300
    a = run(False, n)
    b = run(True, n)
    print("checksums plain / dedup:", a, b, "identical" if a == b else "DIFFERENT")
    ok = a == b
    if "--graphs" in sys.argv:
        c = run(False, n, graphs=True)
        print("checksum plain ring, updates replayed from graphs:", c, "identical" if a == c else "DIFFERENT")
        ok = ok and a == c
    sys.exit(0 if ok else 1)
