#!/usr/bin/env python3
"""Benchmark of the CURL+SAC learner hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one ``CurlSacAgent.update()`` (critic + [actor/alpha + target soft
update on even steps] + CURL, curl_sac.py:426-451) on one per-GPU minibatch of
512 transitions sampled from the HBM-resident replay ring (84x84x9 uint8 frames,
random-crop to 76x76, encoder 4 layers x 32 filters, feature 50, hidden 1024) --
BASELINE.json configs[1]; with N>1 every rank does the same on its own ring
shard and the three gradient buckets are all-reduced over RCCL (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3  # MI355X dense fp32 (vector == f32 MFMA), MI355X_MICROARCH.md chip table
IN_HW, CROP_HW, FRAMES_C, BATCH, HIDDEN = (84, 84), (76, 76), 9, 512, 1024
CAPACITY = 100_000


def conv_layer_flops(hw_in, num_layers=4, nf=32, cin=FRAMES_C):
    """Algorithmic FLOPs per sample of each conv layer (2*Ho*Wo*Cout*Cin*9), SURVEY.md 8d."""
    h, w = (hw_in[0] - 3) // 2 + 1, (hw_in[1] - 3) // 2 + 1
    out = [2.0 * h * w * nf * cin * 9]
    for _ in range(num_layers - 1):
        h, w = h - 2, w - 2
        out.append(2.0 * h * w * nf * nf * 9)
    return out


class NullLogger:
    def log(self, *a, **k):
        pass

    log_histogram = log_param = log_image = log


def cpu_baseline(budget_s=20.0, max_calls=4):
    """The oracle (CPU restatement pinned to the reference) on the host cores:
    a bounded sample of the same workload (same shapes, B=512).  Thread count
    is capped at 32: torch's CPU conv kernels get slower, not faster, beyond
    that on a 256-thread host."""
    from oracle import curla_oracle as O
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    rs = np.random.RandomState(0)
    ag = O.OracleAgent((FRAMES_C,) + CROP_HW, (2,), hidden_dim=HIDDEN)
    B = BATCH

    def batch():
        f = lambda: torch.from_numpy(rs.randint(0, 256, (B, FRAMES_C) + CROP_HW, dtype=np.uint8)).float()  # noqa: E731
        return (f(), torch.from_numpy(rs.uniform(-1, 1, (B, 2)).astype(np.float32)),
                torch.from_numpy(rs.randn(B, 1).astype(np.float32)), f(), torch.ones(B, 1), f(),
                torch.randn(B, 2), torch.randn(B, 2))
    times = []
    t_start = time.perf_counter()
    for s in range(max_calls + 1):  # call 0 is the warm-up unless it alone exhausts the budget
        bt = batch()
        t0 = time.perf_counter()
        ag.update(*bt, step=s)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times
    return {"value": len(timed) / sum(timed), "unit": "batch-512 gradient updates/s", "cores": threads,
            "kind": "port", "sample": f"{len(timed)} OracleAgent.update() call(s) at B=512, 76x76x9, hidden 1024 "
            f"({'after 1 warm-up call' if len(times) > 1 else 'first call, no warm-up: budget exhausted'}; "
            f"{threads} of {os.cpu_count()} host threads), torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--capacity", type=int, default=CAPACITY)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # CURLA_BENCH_FORCE_DIST=1: take the RCCL code path with a single rank (1-GPU check of the N>1 branch)
    distributed = world > 1 or os.environ.get("CURLA_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import curla_amd
    from curla_amd import ops

    # identical parameters on every rank (same seed), rank-specific sampling streams
    curla_amd.set_seed_everywhere(1)
    aug = curla_amd.RandomCrop(IN_HW, CROP_HW)
    agent = curla_amd.CurlSacAgent(
        (FRAMES_C,) + CROP_HW, (2,), dev, aug, hidden_dim=HIDDEN, discount=0.99, init_temperature=0.1, alpha_lr=1e-4,
        alpha_beta=0.5, actor_lr=1e-3, actor_beta=0.9, critic_lr=1e-3, critic_beta=0.9, critic_tau=0.01,
        encoder_feature_dim=50, encoder_lr=1e-3, encoder_tau=0.05, num_layers=4, num_filters=32, log_interval=10 ** 9)
    if distributed:
        agent.enable_data_parallel(single_rank_collectives=(world == 1))
    curla_amd.set_seed_everywhere(1 + rank)

    # replay ring shard, pre-filled on the device with i.i.d. uniform bytes (worst case for any compression)
    cap = args.capacity // world
    rb = curla_amd.ReplayBuffer((FRAMES_C,) + IN_HW, (2,), cap, BATCH, dev, aug)
    g = torch.Generator(device=dev).manual_seed(rank)
    for ring in (rb._obs_store, rb._next_store):
        for s in range(0, ring.numel(), 1 << 28):
            e = min(ring.numel(), s + (1 << 28))
            ring[s:e] = torch.randint(0, 256, (e - s,), dtype=torch.uint8, device=dev, generator=g)
    rb.actions.uniform_(-1, 1, generator=g)
    rb.rewards.normal_(generator=g)
    rb.not_dones.fill_(1.0)
    rb.not_dones[49::50] = 0.0
    rb.idx, rb.full = 0, True

    L = NullLogger()
    # HIP-event timing of the dominant kernel (stride-1 32->32 conv forward) on its own stream
    flops = conv_layer_flops(CROP_HW)
    ev_pairs = []
    real_s1 = ops.conv_s1_fwd
    recording = [False]

    def timed_s1(x, w, b, out):
        if not recording[0]:
            return real_s1(x, w, b, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_s1(x, w, b, out)
        e1.record()
        ev_pairs.append((e0, e1, 2.0 * x.shape[0] * out.shape[1] * out.shape[2] * 32 * 32 * 9))
    ops.conv_s1_fwd = timed_s1

    def barrier():
        if distributed:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # one priming update outside everything (workspaces, split-K buffers and the Adam states are allocated on
    # first use), so that even --warmup 0 times steady-state steps; then the W untimed warm-up steps
    step = 0
    agent.update(rb, L, step)
    step += 1
    # the first ~0.5 s of matrix-pipe work in a process runs ~8 % slow while the clocks ramp: burn it on scratch
    # buffers (no agent state involved) so that short runs (small K and W) also measure the steady state
    bx = torch.zeros((BATCH, 37, 37, 32), device=dev)
    bw, bb = torch.zeros((32, 32, 3, 3), device=dev), torch.zeros(32, device=dev)
    bo = torch.empty((BATCH, 35, 35, 32), device=dev)
    t_burn = time.perf_counter()
    while time.perf_counter() - t_burn < 0.6:
        for _ in range(200):
            real_s1(bx, bw, bb, bo)
        torch.cuda.synchronize()
    del bx, bw, bb, bo
    for _ in range(args.warmup):
        agent.update(rb, L, step)
        step += 1
    barrier()
    # event pairs are taken on a sample of the timed steps spread over the whole region (<= 32 steps = 480
    # launches): thousands of pending HIP events slow the runtime itself and would perturb the measurement
    rec_stride = max(4, args.steps // 32)  # (an instrumented step is ~3 % slower: at most every 4th one)
    t0 = time.perf_counter()
    for i in range(args.steps):
        recording[0] = (i % rec_stride == 0)
        agent.update(rb, L, step)
        step += 1
    recording[0] = False
    barrier()
    dt = time.perf_counter() - t0
    # the timed updates must have produced finite numbers (a NaN run would be meaningless)
    ws = agent._ws(BATCH)
    assert bool(torch.isfinite(ws.scalars).all()) and bool(torch.isfinite(agent._critic_flat).all()), "non-finite state"

    allreduce = None
    if distributed:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # outside the timed region: cost of the three gradient all-reduces of one update, each timed alone
        # (SURVEY.md 8e reporting: all-reduce time per phase and the bus bandwidth it reaches)
        lay = agent._lay
        buckets = {"critic": agent._critic_gflat[lay["enc"][0]:lay["total"]], "actor": agent._actor_gflat,
                   "cpc": agent._critic_gflat[0:lay["enc"][1]]}
        allreduce = {}
        for name, buf in buckets.items():
            scratch = torch.zeros_like(buf)
            for _ in range(3):
                dist.all_reduce(scratch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dist.all_reduce(scratch)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            nbytes = scratch.numel() * 4
            allreduce[name] = {"bytes": nbytes, "ms": ms,
                               "bus_GBps": 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9}

    if rank == 0:
        kflops = sum(p[2] for p in ev_pairs)
        kms = sum(p[0].elapsed_time(p[1]) for p in ev_pairs)
        achieved = kflops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        updates_per_s = world * args.steps / dt
        per_update = BATCH * (5 * sum(flops) + 2 * (2 * sum(flops) - flops[0]))  # n_f=5, n_b=2
        # HBM traffic of the dominant kernel per launch: measured offline with rocprofv3 PMC counters
        # (tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as gfx950 needs),
        # summary committed under profiles/; null if that file is absent
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                traffic = json.load(f)["conv_s1_kernel<0>"]["traffic_bytes"]
        except Exception:
            pass
        # matrix-pipe utilisation of the same kernel from the shader-core PMC pass (tools/pmc_sq.sh), if committed
        mfma_busy = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_sq.json")) as f:
                c = json.load(f)["conv_s1_kernel<0>"]
                mfma_busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
        except Exception:
            pass
        avg_ms = kms / max(1, len(ev_pairs))
        out = {
            "metric": "SAC+CURL gradient updates/sec, batch=512 84x84x9",
            "value": updates_per_s,
            "unit": "batch-512 gradient updates/s (sum over ranks)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: CurlSacAgent.update(), per-GPU batch 512, 84x84x9 uint8 "
                                   "replay ring -> random_crop 76x76, encoder 4x32 filters feat 50, hidden 1024, "
                                   "CURL+critic+actor (actor/target every 2nd step)",
                       "replay_capacity": cap * world, "parallelism": f"dp{world}", "priming_updates": 1, "clock_warmup_s": 0.6},
            "transitions_per_s": updates_per_s * BATCH,
            "conv_algorithmic_gflop_per_update": per_update / 1e9,
            "conv_roofline_frac_whole_update": per_update * (args.steps / dt) / (PEAK_F32_TFLOPS * 1e12),
            "roofline": {"bound": "mfma", "kernel": "conv_s1_kernel<FWD> (3x3 s1 32->32 + bias + ReLU, f32 MFMA 16x16x4)",
                         "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_TFLOPS, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/r01_pmc_traffic.json); "
                                         "algorithmic in+out = 152 MB per launch (mean of the 3 layers)",
                         "launches": len(ev_pairs), "avg_launch_ms": avg_ms,
                         "hbm_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None,
                         "hbm_peak_GBps": 8000.0,
                         "mfma_busy_frac_pmc": mfma_busy},
        }
        if allreduce is not None:
            out["allreduce"] = allreduce
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
