#!/usr/bin/env python3
"""Where a wave's step goes in the bf16x3 stride-1 forward (debug build with cycle stamps):
CURLA_LIB_PATH=tools/_build/libcurla_stamp.so python tools/rwb_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from curla_amd import _lib, ops  # noqa: E402

lib = ctypes.CDLL(_lib.LIB_PATH)
dev = "cuda"
UPDATE = "--update" in sys.argv  # the stamps of the kernel's launches INSIDE update() at configs[1] instead of alone
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
H, L, B1, B2 = 37, 3, 1024, 512
w1 = [r(32, 32, 3, 3) * 0.1 for _ in range(L)]
b1 = [r(32) * 0.1 for _ in range(L)]
x1, x2 = torch.relu(r(B1, H, H, 32)), torch.relu(r(B2, H, H, 32))
mk = lambda B: [torch.empty(B, H - 2 * (i + 1), H - 2 * (i + 1), 32, device=dev) for i in range(L)]  # noqa: E731
o1, o2 = mk(B1), mk(B2)
if UPDATE:
    import curla_amd
    import bench
    curla_amd.set_seed_everywhere(1)
    aug = curla_amd.RandomCrop((84, 84), (76, 76))
    agent = curla_amd.CurlSacAgent((9, 76, 76), (2,), torch.device(dev), aug, hidden_dim=1024, discount=0.99,
                                   init_temperature=0.1, alpha_lr=1e-4, alpha_beta=0.5, critic_tau=0.01, encoder_tau=0.05,
                                   log_interval=10 ** 9)
    rb = curla_amd.ReplayBuffer((9, 84, 84), (2,), 20000, 512, torch.device(dev), aug)
    for ring in (rb._obs_store, rb._next_store):
        ring[:] = torch.randint(0, 256, (ring.numel(),), dtype=torch.uint8, device=dev, generator=g)
    rb.actions.uniform_(-1, 1, generator=g)
    rb.rewards.normal_(generator=g)
    rb.not_dones.fill_(1.0)
    rb.idx, rb.full = 0, True
    L_, step = bench.NullLogger(), [0]

    def run():
        agent.update(rb, L_, step[0])
        step[0] += 1
    n_warm, n_meas = 300, 40
else:
    run = lambda: ops.conv_s1_fwd_stack(x1, w1, b1, o1, x2, w1, b1, o2)  # noqa: E731
    n_warm, n_meas = 200, 20
with _lib.option("s1_fwd", "b3"):
    for _ in range(n_warm):
        run()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.curla_debug_rwb_stamps(out, 1)
    for _ in range(n_meas):
        run()
    torch.cuda.synchronize()
    lib.curla_debug_rwb_stamps(out, 0)
print("inside update() at configs[1] (both stack launches of an update)" if UPDATE else "the critic-phase stack alone, back to back")
n_launch = n_meas * (2 if UPDATE else 1)
print(f"pieces {out[9]} with {out[10]} rows: {out[8] / max(1, out[9]):.0f} cycles per piece from its first load to its last store, "
      f"{out[8] / max(1, out[10]):.0f} per row")
np_ = max(1, out[9])
print("edge steps, cycles each: t=0 %.0f  t=1 %.0f  t=n %.0f  t=n+1 %.0f" % tuple(out[11 + e] / np_ for e in range(4)))
n = max(1, out[3])
print(f"whole kernel, wave 0 of workgroup 7: {out[15]} shader cycles in {out[14]} ticks of the 100 MHz clock: "
      f"{out[15] / max(1, out[14]) * 0.1:.2f} GHz, {out[14] / n_launch / 100:.1f} us per launch")
nl = max(1, out[7])
print(f"layers {nl}: filter build + barrier {out[4] / nl:.0f}  run_layer {out[5] / nl:.0f}  end-of-layer barrier {out[6] / nl:.0f} cycles per layer; "
      f"sum over a launch's 3 layers {(out[4] + out[5] + out[6]) / nl * 3:.0f}")
print(f"full steps {n}: split phase {out[0] / n:.0f}  product phase {out[1] / n:.0f}  finish {out[2] / n:.0f} cycles per step "
      f"(s_memtime ticks; 144 matrix instructions = 2304 cycles of the matrix pipe)")
