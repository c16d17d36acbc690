#!/usr/bin/env python3
"""The dominant kernel's clock and matrix-pipe utilisation from the SAME launches, three ways.

Runs conv_rwb_fwd_kernel from a diagnostic build (``-DRWB_CLOCK``: two stamps per workgroup and launch, nothing in the
loops) and reports, per launch (median over the 256 workgroups):
  * shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) from a workgroup's first instruction to its last
    -> the in-kernel clock and the kernel's duration as the waves themselves see them;
  * the matrix instructions of the launch (counted from the shapes, checked against SQ_INSTS_MFMA when a PMC pass
    supplies it) x 16 cycles / (1024 SIMDs x elapsed shader cycles) = matrix-pipe utilisation at the clock the chip held.
``--mode stack``: configs[1]'s critic-phase stack alone, back to back (the s1_bench / rwb_stamps setting);
``--mode update``: inside real ``update()`` calls at configs[1] (both launch kinds of an update: the setting bench.py
measures).  tools/clock_reconcile.sh runs both, plain and under rocprofv3 --pmc, and puts the counters beside them."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import curla_amd  # noqa: E402
from curla_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=("stack", "update"), default="stack")
ap.add_argument("--settle-s", type=float, default=2.0, help="seconds of back-to-back work before the measured launches")
ap.add_argument("--launches", type=int, default=60)
ap.add_argument("--out", default=None)
args = ap.parse_args()

lib = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(lib, "curla_debug_rwb_clock"):
    sys.exit("needs the diagnostic build: tools/build_variant.sh clock -DRWB_CLOCK; CURLA_LIB_PATH=tools/_build/libcurla_clock.so")
dev = torch.device("cuda")
NWG = torch.cuda.get_device_properties(dev).multi_processor_count


def mfma_per_launch(samples_hw):
    """Matrix instructions of a stack launch: 144 per step of a wave (4 Winograd positions x 3 row taps x 2 channel
    halves x 6 bf16 products), a step = one output row of 16 pixel-pair columns; the pair columns a row has beyond a
    multiple of 16 are cut into vertical segments (conv_rwb.h), so a sample's layer takes ceil(Ho x ceil(Wo / 2) / 16)
    steps -- 40 + 36 + 31 for configs[1]'s 35 / 33 / 31 rows (SQ_INSTS_MFMA confirms it: tools/clock_reconcile.sh)."""
    n = 0
    for B, H, W, layers in samples_hw:
        h, w = H, W
        for _ in range(layers):
            h, w = h - 2, w - 2
            n += B * (-(-(h * ((w + 1) // 2)) // 16)) * 144
    return n


if args.mode == "stack":
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    H, L, B1, B2 = 37, 3, 1024, 512
    w1 = [r(32, 32, 3, 3) * 0.1 for _ in range(L)]
    b1 = [r(32) * 0.1 for _ in range(L)]
    x1, x2 = torch.relu(r(B1, H, H, 32)), torch.relu(r(B2, H, H, 32))
    mk = lambda B: [torch.empty(B, H - 2 * (i + 1), H - 2 * (i + 1), 32, device=dev) for i in range(L)]  # noqa: E731
    o1, o2 = mk(B1), mk(B2)
    run = lambda: ops.conv_s1_fwd_stack(x1, w1, b1, o1, x2, w1, b1, o2)  # noqa: E731
    per_call, mfma = 1, mfma_per_launch([(B1 + B2, H, H, L)])
    what = "configs[1] critic-phase stack alone (1024 + 512 samples x 3 layers from 37x37), back to back"
else:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench  # noqa: E402
    cfg = bench.CONFIGS["c2"]
    curla_amd.set_seed_everywhere(1)
    aug = curla_amd.RandomCrop((84, 84), (76, 76))
    agent = curla_amd.CurlSacAgent((9, 76, 76), (2,), dev, aug, hidden_dim=1024, discount=0.99, init_temperature=0.1,
                                   alpha_lr=1e-4, alpha_beta=0.5, critic_tau=0.01, encoder_tau=0.05, log_interval=10 ** 9)
    rb = curla_amd.ReplayBuffer((9, 84, 84), (2,), 20000, 512, dev, aug)
    gg = torch.Generator(device=dev).manual_seed(0)
    for ring in (rb._obs_store, rb._next_store):
        ring[:] = torch.randint(0, 256, (ring.numel(),), dtype=torch.uint8, device=dev, generator=gg)
    rb.actions.uniform_(-1, 1, generator=gg)
    rb.rewards.normal_(generator=gg)
    rb.not_dones.fill_(1.0)
    rb.idx, rb.full = 0, True
    L_ = bench.NullLogger()
    step = [0]

    def run():
        agent.update(rb, L_, step[0])
        step[0] += 1
    per_call = 2  # stack launches per update: critic phase (1024 + 512 samples) and actor / CURL phase (512 + 512)
    mfma = (mfma_per_launch([(1536, 37, 37, 3)]) + mfma_per_launch([(1024, 37, 37, 3)])) / 2.0
    what = "inside update() at configs[1] (B = 512): both stack launches of an update, averaged"

with _lib.option("s1_fwd", "b3"):
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.settle_s:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    lib.curla_debug_rwb_clock(None, NWG, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.launches):
        run()
    e1.record()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * (3 * NWG))()
    assert lib.curla_debug_rwb_clock(out, NWG, 0) == 0
a = np.array(list(out), dtype=np.float64).reshape(NWG, 3)
a = a[a[:, 2] > 0]
n_launch = a[:, 2]
assert np.all(n_launch == args.launches * per_call), (n_launch.min(), n_launch.max(), args.launches * per_call)
cyc, tick = a[:, 0] / n_launch, a[:, 1] / n_launch
clock = cyc / tick * 0.1  # GHz
res = dict(mode=args.mode, what=what, launches=int(args.launches * per_call), workgroups=int(len(a)),
           shader_cycles_per_launch=dict(median=float(np.median(cyc)), min=float(cyc.min()), max=float(cyc.max())),
           kernel_us_in_kernel=dict(median=float(np.median(tick)) / 100.0, max=float(tick.max()) / 100.0),
           in_kernel_clock_GHz=dict(median=float(np.median(clock)), min=float(clock.min()), max=float(clock.max())),
           mfma_per_launch_from_shapes=float(mfma),
           wall_ms_per_call_hip_events=e0.elapsed_time(e1) / args.launches)
# utilisation at the clock the chip held: the longest-living workgroup bounds the launch
res["matrix_pipe_util_at_in_kernel_clock"] = mfma * 16.0 / (4.0 * NWG * float(cyc.max()))
res["issued_TFLOPs"] = mfma * 16384.0 / (float(tick.max()) * 1e-8) / 1e12
res["frac_of_2500_TF_nominal_peak"] = res["issued_TFLOPs"] / 2500.0
print(json.dumps(res))
if args.out:
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
