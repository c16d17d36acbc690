#!/usr/bin/env python3
"""Benchmark of the CURL+SAC learner hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c5] [--no-others] [--dry-run]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one ``CurlSacAgent.update()`` (critic + [actor/alpha + target soft
update on even steps] + CURL, curl_sac.py:426-451) on one per-GPU minibatch
sampled from the HBM-resident replay ring.  ``--config`` picks the BASELINE.json
configuration (default c2 = configs[1], the one the metric is quoted on):
  c2  B=512, 84x84x9 uint8 ring -> random_crop 76x76, 4 conv layers, CURL+critic+actor
  c3  --pixel_sac: identity augmentation (84x84 un-cropped, train.py:262-264), no CURL head, B=512
  c5  B=1024, 168x168x12 (frame_stack 4), color_jiggle, 6 conv layers (configs[4] per GPU)
  c1t the reference AS SHIPPED (train.py:45-46,72; augmentations.py:21-24; encoder.py:26,42-43): B=512, 90x160x9 uint8
      ring -> the default RandomCrop (factor 0.84: 76x135), 4 conv layers (31x61 out, fc over 60512), CURL+critic+actor
With N>1 every rank does the same on its own ring shard and the gradient buckets
are all-reduced over RCCL (weak scaling).  Prints ONE JSON line on rank 0.

Without ``--config`` the line's ``metric`` / ``value`` / ``config`` are c2's and, at N = 1, the same process then
measures c1t, c3 and c5 as well (the c2 ring is freed first) and reports them under ``other_configs`` -- each with its own
``value``, ``ms_per_step``, ``steps``, ``roofline`` and ``cpu_baseline``; ``--no-others`` or an explicit ``--config``
measures one configuration only.  At N > 1 only c2 is measured unless ``--others`` asks for all three (a failure in a
secondary configuration of a multi-GPU job would take the headline line down with it).

At N > 1 BOTH data-parallel schedules are measured (``--schedule auto``, the default): the blocking one (one all-reduce
per gradient bucket in front of its optimizer step) and the overlapped one (each bucket in two asynchronous pieces under
the conv backward); the line's ``value`` is the faster one's K timed steps, ``schedule`` names it and ``allreduce`` holds
both (``blocking`` / ``overlapped``: value, ms_per_step, communication exposed per step) next to the per-bucket
all-reduce times and bus bandwidths.  ``--schedule blocking|overlapped`` measures one.

``--sweep``: ONE job of N ranks measures the sub-groups 1, 2, 4, ... N in turn (ranks 0 .. n-1 compute, the others wait
on a CPU barrier), both schedules each, and prints one JSON line per n plus a summary line with the scaling efficiency
per n and the schedule the measurement picks -- the whole multi-GPU decision from one command on one node:
    python bench.py --gpus 8 --sweep [--others]

``--dry-run`` walks the same host path -- launcher respawn, ring shards, rank seeds, data-parallel setup, the
update loop, the all-reduce report, the one JSON line -- with gloo on the CPU and the kernel calls routed to the
launch-trace hook (nothing is computed, no GPU is touched): the N>1 plumbing can be rehearsed without a node
(tests/test_bench_dry_run.py).  Its numbers mean nothing and the line says ``"dry_run": true``.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3  # MI355X dense fp32 (vector == f32 MFMA), MI355X_MICROARCH.md chip table
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (no sparsity), MI355X_MICROARCH.md chip table
HIDDEN, CAPACITY = 1024, 100_000

CONFIGS = {
    "c2": dict(baseline_index=1, obs=(9, 84, 84), crop=(76, 76), aug="random_crop", layers=4, batch=512, pixel_sac=False,
               metric="SAC+CURL gradient updates/sec, batch=512 84x84x9",
               workload="BASELINE.json configs[1]: CurlSacAgent.update(), per-GPU batch 512, 84x84x9 uint8 replay ring "
                        "-> random_crop 76x76, encoder 4x32 filters feat 50, hidden 1024, CURL+critic+actor "
                        "(actor/target every 2nd step)", cpu_batch=512, cpu_calls=4),
    "c3": dict(baseline_index=2, obs=(9, 84, 84), crop=None, aug="identity", layers=4, batch=512, pixel_sac=True,
               metric="SAC (--pixel_sac) gradient updates/sec, batch=512 84x84x9",
               workload="BASELINE.json configs[2]: CurlSacAgent.update() with pixel_sac=True, per-GPU batch 512, "
                        "84x84x9 uint8 replay ring, identity augmentation (84x84 un-cropped, train.py:262-264), encoder "
                        "4x32 filters feat 50, hidden 1024, critic+actor only (actor/target every 2nd step)",
               cpu_batch=512, cpu_calls=4),
    "c5": dict(baseline_index=4, obs=(12, 168, 168), crop=None, aug="color_jiggle", layers=6, batch=1024,
               pixel_sac=False, metric="SAC+CURL gradient updates/sec, batch=1024 168x168x12 color_jiggle L=6",
               workload="BASELINE.json configs[4] (one GPU's share): CurlSacAgent.update(), per-GPU batch 1024, "
                        "168x168x12 uint8 replay ring (frame_stack 4), color_jiggle augmentation, encoder 6x32 filters "
                        "feat 50, hidden 1024, CURL+critic+actor (actor/target every 2nd step)",
               cpu_batch=32, cpu_calls=2),
    # not a BASELINE.json configuration: the one geometry the unmodified reference runs (its encoder's shape table and
    # its RandomCrop's factor admit nothing else with 90x160 frames), at train.py's batch size -- reported beside them
    "c1t": dict(baseline_index=None, obs=(9, 90, 160), crop=(76, 135), aug="random_crop_default", layers=4, batch=512,
                pixel_sac=False, metric="SAC+CURL gradient updates/sec, batch=512 90x160x9 -> 76x135 (reference as shipped)",
                workload="the reference's shipped configuration (train.py:45-46,72; augmentations.py:21-24; encoder.py:26,"
                         "42-43): CurlSacAgent.update(), per-GPU batch 512, 90x160x9 uint8 replay ring -> default RandomCrop "
                         "(ceil(0.84 x side) = 76x135), encoder 4x32 filters (31x61 out) feat 50, hidden 1024, "
                         "CURL+critic+actor (actor/target every 2nd step)", cpu_batch=128, cpu_calls=3),
}


def conv_layer_flops(hw_in, num_layers, cin, nf=32):
    """Algorithmic FLOPs per sample of each conv layer (2*Ho*Wo*Cout*Cin*9), SURVEY.md 8d."""
    h, w = (hw_in[0] - 3) // 2 + 1, (hw_in[1] - 3) // 2 + 1
    out = [2.0 * h * w * nf * cin * 9]
    for _ in range(num_layers - 1):
        h, w = h - 2, w - 2
        out.append(2.0 * h * w * nf * nf * 9)
    return out


def flops_per_update(cfg, step):
    """CURL mode n_f=5, n_b=2 on every step; pixel_sac n_f=4 (even) / 3 (odd), n_b=1."""
    hw = cfg["crop"] or cfg["obs"][1:]
    f = conv_layer_flops(hw, cfg["layers"], cfg["obs"][0])
    fc, f1 = sum(f), f[0]
    if cfg["pixel_sac"]:
        return cfg["batch"] * ((4 if step % 2 == 0 else 3) * fc + (2 * fc - f1))
    return cfg["batch"] * (5 * fc + 2 * (2 * fc - f1))


def matrix_pipe_seconds_per_update(cfg, step, opts):
    """Seconds per update the matrix pipes would need AT THEIR DENSE PEAKS for the conv work as the kernels issue it
    (advisor, round 5: the fp32-roofline fractions above are not bounded by 1 once the stride-1 convs run on the bf16
    matrix cores).  bf16 pipe (2.5 PFLOP/s): stride-1 forward and data gradient = direct FLOPs / 1.5 (Winograd F(2,3)
    along x) x 6 products (bf16x3); the uint8 first layer and its weight gradient = direct FLOPs x 3 products (a pixel is
    exact in one part).  f32-input pipe (157.3 TFLOP/s): the stride-1 weight gradient = direct / 2.25 (F(3,2) in both
    directions; / 1.5 along x only), anything the options put back on that pipe, and the float first layer of the
    colour-jittered configuration.  Strip padding and the k-padding of the first layer (27 -> 32) are not counted."""
    hw = cfg["crop"] or cfg["obs"][1:]
    f = conv_layer_flops(hw, cfg["layers"], cfg["obs"][0])
    f1, fs = f[0], sum(f[1:])
    n_f, n_b = ((4 if step % 2 == 0 else 3), 1) if cfg["pixel_sac"] else (5, 2)
    bf16 = f32 = 0.0
    s1 = opts.get("s1_fwd", "auto")
    if s1 in ("auto", "b3"):
        bf16 += (n_f + n_b) * fs / 1.5 * 6
    else:
        f32 += (n_f + n_b) * fs / (2.0 if s1 == "f43" else 1.5)
    f32 += n_b * fs / (1.5 if opts.get("s1_wgrad", "auto") == "x" else 2.25)
    u8 = cfg["aug"] != "color_jiggle"
    if u8 and opts.get("conv1_u8", "auto") in ("auto", "rwb") and 3 * cfg["obs"][0] <= 32:
        bf16 += n_f * f1 * 3
    else:
        f32 += n_f * f1
    if u8 and opts.get("wgrad1_u8", "auto") in ("auto", "b16") and 3 * cfg["obs"][0] <= 32:
        bf16 += n_b * f1 * 3
    else:
        f32 += n_b * f1
    return cfg["batch"] * (bf16 / (PEAK_BF16_TFLOPS * 1e12) + f32 / (PEAK_F32_TFLOPS * 1e12))


def committed_clock(kernel):
    """The in-kernel clock and matrix-pipe utilisation of the dominant kernel INSIDE update() at configs[1], measured
    offline by tools/clock_reconcile.sh (diagnostic build with two stamps per workgroup; the PMC counters of the same
    launches are in the .txt beside it) -- quoted while the summary's source hash is the built library's."""
    from curla_amd import build
    try:
        with open(os.path.join(ROOT, "profiles", "r06_clock_reconcile.json")) as f:
            d = json.load(f)
    except Exception:
        return None
    if d.get("_source_hash") != build.source_hash() or d.get("kernel") != kernel or "update" not in d:
        return None
    u = d["update"]
    return {"in_kernel_clock_GHz": u["in_kernel_clock_GHz"], "matrix_pipe_busy_at_that_clock": u["matrix_pipe_util_at_in_kernel_clock"],
            "how": "s_memtime / s_memrealtime stamps of every workgroup's first and last instruction, conv_rwb_fwd_kernel's "
                   "launches inside update() at configs[1]; matrix instructions x 16 cycles / (1024 SIMDs x elapsed shader "
                   "cycles); profiles/r06_clock_reconcile.txt has the PMC counters of the same launches "
                   "(SQ_BUSY_CU_CYCLES / 256 and SQ_WAVE_CYCLES x 4 / 2048 agree with the stamps to 1-3 %)"}


class NullLogger:
    def log(self, *a, **k):
        pass

    log_histogram = log_param = log_image = log


def _oracle_sample(cfg, batch, calls, budget_s, threads):
    """``calls`` timed OracleAgent.update() calls at batch ``batch`` after one warm-up call (stops early when the
    time budget is spent).  Returns (seconds per call, calls timed, had_warmup)."""
    from oracle import curla_oracle as O
    rs = np.random.RandomState(0)
    shape = (cfg["obs"][0],) + tuple(cfg["crop"] or cfg["obs"][1:])
    ag = O.OracleAgent(shape, (2,), hidden_dim=HIDDEN, num_layers=cfg["layers"], pixel_sac=cfg["pixel_sac"])

    def minibatch():
        f = lambda: torch.from_numpy(rs.randint(0, 256, (batch,) + shape, dtype=np.uint8)).float()  # noqa: E731
        return (f(), torch.from_numpy(rs.uniform(-1, 1, (batch, 2)).astype(np.float32)),
                torch.from_numpy(rs.randn(batch, 1).astype(np.float32)), f(), torch.ones(batch, 1), f(),
                torch.randn(batch, 2), torch.randn(batch, 2))
    times = []
    t_start = time.perf_counter()
    for s in range(calls + 1):  # call 0 is the warm-up unless it alone exhausts the budget
        bt = minibatch()
        t0 = time.perf_counter()
        ag.update(*bt, step=s)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times
    return sum(timed) / len(timed), len(timed), len(times) > 1


def cpu_baseline(cfg, name):
    """The oracle (CPU restatement pinned to the reference) on the host cores, on a bounded sample of the same
    workload.  Thread count is capped at 32: torch's CPU conv kernels get slower, not faster, beyond that on a
    256-thread host.  When the sample's batch is smaller than the configuration's, the rate is scaled to the
    metric's unit (updates of the full batch per second) and the sample says so.  For c2 the reference's own
    CPU-runnable case, BASELINE.json configs[0] (B=32), is timed as well."""
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    B, b = cfg["batch"], cfg["cpu_batch"]
    sec, n, warm = _oracle_sample(cfg, b, cfg["cpu_calls"], 25.0, threads)
    shape = (cfg["obs"][0],) + tuple(cfg["crop"] or cfg["obs"][1:])
    out = {"value": (1.0 / sec) * b / B, "unit": f"batch-{B} gradient updates/s", "cores": threads, "kind": "port",
           "sample": f"{n} OracleAgent.update() call(s) at B={b}, {shape[1]}x{shape[2]}x{shape[0]}, "
                     f"{cfg['layers']} conv layers, hidden {HIDDEN}{', pixel_sac' if cfg['pixel_sac'] else ''} "
                     f"({'after 1 warm-up call' if warm else 'first call, no warm-up: budget exhausted'}; "
                     f"{threads} of {os.cpu_count()} host threads), torch {torch.__version__} CPU"
                     + ("" if b == B else f"; rate scaled by {b}/{B} to the batch-{B} unit"),
           "transitions_per_s": b / sec}
    if name == "c2":
        sec0, n0, _ = _oracle_sample(cfg, 32, 200, 25.0, threads)
        out["configs0"] = {"value": 1.0 / sec0, "unit": "batch-32 gradient updates/s", "calls": n0,
                           "note": "BASELINE.json configs[0] shapes (B=32, 76x76 crop of 84x84x9) on the oracle; the "
                                   "reference itself measured 9.30/s on 8 vCPUs in the build container (SURVEY.md 6)"}
    return out


def committed_counters(cfg_name, kernel):
    """HBM bytes per launch and matrix-pipe busy fraction of ``kernel`` from the committed rocprofv3 PMC summaries
    (tools/profile_config.sh + tools/pmc_summarize.py: separate --pmc passes, FETCH_SIZE doubled as gfx950 needs).
    They are measured offline, so they are only quoted when the summary was taken from the kernel sources this
    library was built from (the summary records their hash); otherwise null.  The newest round's summary wins."""
    from curla_amd import build
    cur = build.source_hash()
    traffic = busy = None
    note = "no committed PMC summary for this configuration"
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        for fn, key in ((f"{rnd}_pmc_traffic_{cfg_name}.json", "traffic"), (f"{rnd}_pmc_sq_{cfg_name}.json", "sq")):
            try:
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    d = json.load(f)
            except Exception:
                continue
            if d.get("_source_hash") != cur:
                if traffic is None and busy is None:
                    note = (f"profiles/{fn} was measured on other kernel sources ({d.get('_source_hash')} != {cur}): "
                            "not quoted")
                continue
            if kernel not in d:
                continue
            if key == "traffic" and traffic is None:
                traffic = d[kernel]["traffic_bytes"]
                note = f"profiles/{fn} (rocprofv3 --pmc, same kernel sources {cur})"
            elif key == "sq" and busy is None:
                c = d[kernel]
                busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
    return traffic, busy, note


def live_counters(cfg_name, kernel, timeout_s=150):
    """HBM traffic and matrix-pipe busy fraction of ``kernel`` measured NOW, by this run: three short child runs of this
    script under ``rocprofv3 --pmc`` (separate passes for FETCH_SIZE, WRITE_SIZE and the two SQ counters, as
    MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled as gfx950 needs: tools/pmc_summarize.py), each a fresh process --
    the children are started, never exec'd into, and this process holds no HIP state they need.  Returns (traffic bytes
    per launch, busy fraction, note) or None when rocprofv3 is missing, a pass fails or the budget runs out: the
    caller then quotes the committed summaries (committed_counters)."""
    import csv
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    t_end = time.time() + timeout_s
    out = tempfile.mkdtemp(prefix="curla_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for tag, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                              ("sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"])):
            left = t_end - time.time()
            if left < 20:
                return None
            cmd = [rp, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", os.path.join(out, tag), "--",
                                               sys.executable, os.path.abspath(__file__), "--config", cfg_name, "--steps", "6",
                                               "--warmup", "2", "--no-cpu-baseline", "--clock-warmup-s", "0", "--capacity",
                                               "20000", "--no-live-pmc"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left)
            if r.returncode != 0:
                return None
            for f in glob.glob(os.path.join(out, tag, "*", "*counter_collection.csv")):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if kernel in row["Kernel_Name"]:
                            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        mean = lambda k: sum(vals[k]) / len(vals[k])  # noqa: E731
        traffic = 2.0 * mean("FETCH_SIZE") * 1024.0 + mean("WRITE_SIZE") * 1024.0
        busy = mean("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * mean("SQ_BUSY_CU_CYCLES"))
        return traffic, busy, (f"measured by this run: rocprofv3 --pmc child runs of this script (6 updates each; FETCH_SIZE x 2 "
                               f"+ WRITE_SIZE, KiB; means over {len(vals['FETCH_SIZE'])} launches)")
    except Exception:  # noqa: BLE001  (a missing counter, a timeout, a profiler that is not allowed here ...)
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child job (before anything here has
    touched the GPU) and leave with its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


class Job:
    """What one rank knows about the run: who it is, where it computes, how it synchronises and measures time."""

    def __init__(self, args):
        self.args = args
        self.dry = args.dry_run
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={self.world} ranks")
        # CURLA_BENCH_FORCE_DIST=1: take the RCCL code path with a single rank (1-GPU check of the N>1 branch)
        self.distributed = self.world > 1 or os.environ.get("CURLA_BENCH_FORCE_DIST") == "1"
        if self.dry:
            from curla_amd import _lib
            self.dev = torch.device("cpu")
            self.launches = 0

            def hook(name, a):
                self.launches += 1
            _lib.set_trace_hook(hook)
            torch.set_num_threads(2)
        else:
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        if self.distributed:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if self.dry:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
        self.ctl = None  # (sweep: a gloo group for the barriers the idle ranks wait at -- no kernel spins on their GPUs)

    def sync(self):
        if not self.dry:
            torch.cuda.synchronize()

    def barrier(self, pg=None, n=None):
        """Barrier over the ranks of ``pg`` (default: all) + device sync.  ``n == 1``: nobody to wait for."""
        if self.distributed and (n is None or n > 1 or pg is not None):
            import torch.distributed as dist
            dist.barrier(group=pg)
        self.sync()

    def host_barrier(self):
        """All ranks, on the CPU (the sweep's idle ranks wait here while a sub-group measures)."""
        self.sync()
        if self.ctl is not None:
            import torch.distributed as dist
            dist.barrier(group=self.ctl)

    def cu_count(self):
        return 256 if self.dry else torch.cuda.get_device_properties(self.dev).multi_processor_count

    def timer(self):
        """(start, stop -> ms): HIP events on the current stream; the host clock in a dry run."""
        if self.dry:
            t = [0.0]
            return (lambda: t.__setitem__(0, time.perf_counter())), (lambda: 1e3 * (time.perf_counter() - t[0]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def stop():
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1)
        return e0.record, stop

    def close(self):
        if self.distributed:
            import torch.distributed as dist
            dist.destroy_process_group()
        if self.dry:
            from curla_amd import _lib
            _lib.set_trace_hook(None)


def measure(job, name, steps, warmup, with_cpu_baseline, pg=None, n=None, overlap=None):
    """One configuration: build agent + ring shard, prime, warm up, time ``steps`` updates between barriers.
    Returns the result object on rank 0 (None elsewhere).  Everything it allocated is released on return.
    ``pg`` / ``n``: the process group and the number of ranks this measurement runs on (default: all of the job's;
    the sweep passes ranks 0 .. n-1 -- only they call); ``overlap``: the data-parallel schedule (None: the agent's
    default, CURLA_DP_OVERLAP)."""
    import curla_amd
    from curla_amd import ops
    args, cfg, dev, rank = job.args, CONFIGS[name], job.dev, job.rank
    world = job.world if n is None else n
    distributed = job.distributed and (world > 1 or os.environ.get("CURLA_BENCH_FORCE_DIST") == "1")
    launches0 = job.launches if job.dry else 0

    C, (H, W), B = cfg["obs"][0], cfg["obs"][1:], cfg["batch"]
    curla_amd.set_seed_everywhere(1)
    from curla_amd import augmentations as A  # (make_augmentor prints a banner; stdout carries the JSON line only)
    aug = {"random_crop": lambda: A.RandomCrop((H, W), cfg["crop"]), "identity": lambda: A.IdentityAugmentation((H, W)),
           "random_crop_default": lambda: A.RandomCrop((H, W)),  # (the reference's own factor decides the crop)
           "color_jiggle": lambda: A.ColorJiggle((H, W))}[cfg["aug"]]()
    assert cfg["crop"] is None or tuple(aug.output_shape) == tuple(cfg["crop"]), (aug.output_shape, cfg["crop"])
    agent = curla_amd.CurlSacAgent(
        (C,) + tuple(aug.output_shape), (2,), dev, aug, hidden_dim=HIDDEN, discount=0.99, init_temperature=0.1,
        alpha_lr=1e-4, alpha_beta=0.5, actor_lr=1e-3, actor_beta=0.9, critic_lr=1e-3, critic_beta=0.9, critic_tau=0.01,
        encoder_feature_dim=50, encoder_lr=1e-3, encoder_tau=0.05, num_layers=cfg["layers"], num_filters=32,
        pixel_sac=cfg["pixel_sac"], log_interval=10 ** 9)
    if distributed:
        # rank 0's parameters are broadcast
        agent.enable_data_parallel(process_group=pg, single_rank_collectives=(world == 1), overlap=overlap)
    seed = 1 + rank
    curla_amd.set_seed_everywhere(seed)  # rank-specific sampling / policy-noise streams

    # replay ring shard.  SURVEY.md 8d recipe: obs, next_obs i.i.d. uniform bytes (worst case for any compression),
    # action ~ U(-1,1)^2, reward ~ N(0,1), not_done = 1 except every 50th transition.
    cap = args.capacity // world
    rb = curla_amd.ReplayBuffer((C, H, W), (2,), cap, B, dev, aug)
    if args.prefill == "host":
        # the recipe to the letter: RandomState(0) on the host, transition by transition (slow: ~1 GB/s)
        rs = np.random.RandomState(rank)
        for s in range(0, cap, 512):
            n = min(512, cap - s)
            rb.add_batch(rs.randint(0, 256, (n, C, H, W), dtype=np.uint8), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                         rs.randn(n).astype(np.float32), rs.randint(0, 256, (n, C, H, W), dtype=np.uint8),
                         (np.arange(s, s + n) % 50) == 49)
        prefill = "SURVEY.md 8d recipe on the host: RandomState(rank) uniform bytes, U(-1,1) actions, N(0,1) rewards"
    else:
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
        prefill = ("same distributions as the SURVEY.md 8d recipe (uniform bytes, U(-1,1) actions, N(0,1) rewards, "
                   "every 50th transition terminal) drawn by a device generator seeded with the rank instead of "
                   "numpy RandomState(0) on the host (--prefill host follows the recipe to the letter)")

    L = NullLogger()
    # HIP-event timing of the dominant kernel (stride-1 32->32 conv forward) on the stream it is launched on
    ev_pairs = []
    recording = [False]
    real = (ops.conv_s1_fwd, ops.conv_s1_fwd2, ops.conv_s1_fwd_stack)

    def timed(fn, flops, nbytes, *a):
        if not recording[0] or job.dry:
            return fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*a)
        e1.record()
        ev_pairs.append((e0, e1, flops, nbytes))

    def timed_s1(x, w, b, out):
        return timed(real[0], 2.0 * x.shape[0] * out.shape[1] * out.shape[2] * 32 * 32 * 9,
                     4.0 * (x.numel() + out.numel()), x, w, b, out)

    def timed_s1_2(x, w, b, out, x2, w2, b2, out2):  # two minibatches (own weights each) in one launch
        return timed(real[1], 2.0 * (x.shape[0] + x2.shape[0]) * out.shape[1] * out.shape[2] * 32 * 32 * 9,
                     4.0 * (x.numel() + out.numel() + x2.numel() + out2.numel()), x, w, b, out, x2, w2, b2, out2)

    def timed_stack(x, ws, bs, outs, x2=None, ws2=None, bs2=None, outs2=None):  # all stride-1 layers, one launch
        if not recording[0] or job.dry:
            return real[2](x, ws, bs, outs, x2, ws2, bs2, outs2)
        fl = by = 0.0
        for (xin, os_) in ((x, outs), (x2, outs2)):
            if xin is None:
                continue
            prev = xin
            for o in os_:
                fl += 2.0 * prev.shape[0] * o.shape[1] * o.shape[2] * 32 * 32 * 9
                by += 4.0 * (prev.numel() + o.numel())
                prev = o
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ok = real[2](x, ws, bs, outs, x2, ws2, bs2, outs2)
        e1.record()
        if ok:
            ev_pairs.append((e0, e1, fl, by))
        return ok
    ops.conv_s1_fwd, ops.conv_s1_fwd2, ops.conv_s1_fwd_stack = timed_s1, timed_s1_2, timed_stack
    try:
        # one priming update outside everything (workspaces, split-K buffers and the Adam states are allocated on
        # first use), so that even --warmup 0 times steady-state steps; then the W untimed warm-up steps
        step = 0
        agent.update(rb, L, step)
        step += 1
        # the first ~0.5 s of matrix-pipe work in a process runs ~8 % slow while the clocks ramp: burn it on scratch
        # buffers (no agent state involved) so that short runs (small K and W) also measure the steady state.  The
        # burn uses a kernel instance no configuration launches (first-layer conv from a float NCHW tensor with
        # C = 3), so the rocprofv3 per-kernel averages of this command contain the updates' launches only.
        if not job.dry and args.clock_warmup_s > 0:
            bx = ops.ObsRef.from_tensor(torch.zeros((512, 3, 84, 84), device=dev))
            bw, bb = torch.zeros((32, 3, 3, 3), device=dev), torch.zeros(32, device=dev)
            bo = torch.empty((512, 41, 41, 32), device=dev)
            t_burn = time.perf_counter()
            while time.perf_counter() - t_burn < args.clock_warmup_s:
                for _ in range(200):
                    ops.conv1_fwd(bx, bw, bb, bo)
                torch.cuda.synchronize()
            del bx, bw, bb, bo
        graphed = bool(args.graphs and not distributed and not job.dry and rb.graph_supported())
        if graphed:
            agent.enable_update_graphs(rb)
            for _ in range(6):  # one eager warm-up + two captures per kind, outside the W warm-up steps
                agent.update(rb, L, step)
                step += 1
        for _ in range(warmup):
            agent.update(rb, L, step)
            step += 1
        job.barrier(pg, world)
        # event pairs are taken on a sample of the timed steps spread over the whole region (<= 32 steps): thousands
        # of pending HIP events slow the runtime itself and would perturb the measurement
        rec_stride = max(4, steps // 32)  # (an instrumented step is ~3 % slower: at most every 4th one)
        first_step = step
        t0 = time.perf_counter()
        for i in range(steps):
            recording[0] = (i % rec_stride == 0)
            agent.update(rb, L, step)
            step += 1
        recording[0] = False
        job.barrier(pg, world)
        dt = time.perf_counter() - t0
        if graphed:  # a replayed update makes no host-side launch to put events around: time the kernel on eager updates
            agent.disable_update_graphs()
            recording[0] = True
            for _ in range(8):
                agent.update(rb, L, step)
                step += 1
            recording[0] = False
            job.sync()
    finally:
        ops.conv_s1_fwd, ops.conv_s1_fwd2, ops.conv_s1_fwd_stack = real
    # the timed updates must have produced finite numbers (a NaN run would be meaningless)
    if not job.dry:
        ws = agent._ws(B)
        assert bool(torch.isfinite(ws.scalars).all()) and bool(torch.isfinite(agent._critic_flat).all()), \
            "non-finite state"

    allreduce = None
    if distributed:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=pg)
        dt = float(t.item())
        # outside the timed region: cost of the gradient all-reduces of one update, each bucket timed alone
        # (SURVEY.md 8e reporting: all-reduce time per phase and the bus bandwidth it reaches)
        lay = agent._lay
        buckets = {"critic": agent._critic_gflat[lay["enc"][0]:lay["total"]], "actor": agent._actor_gbucket}
        if not cfg["pixel_sac"]:
            buckets["cpc"] = agent._critic_gflat[0:lay["enc"][1]]
        allreduce = {"overlapped_with_backward": bool(agent._dp_overlap), "buckets": {}}
        reps = 2 if job.dry else 10
        for bname, buf in buckets.items():
            scratch = torch.zeros_like(buf)
            for _ in range(1 if job.dry else 3):
                dist.all_reduce(scratch, group=pg)
            job.sync()
            start, stop = job.timer()
            start()
            for _ in range(reps):
                dist.all_reduce(scratch, group=pg)
            ms = stop() / reps
            nbytes = scratch.numel() * 4
            allreduce["buckets"][bname] = {"bytes": nbytes, "ms": ms,
                                           "bus_GBps": 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9}
        # every rank's shard size and sampling seed, for the line (rank 0 reports what the ranks actually used)
        mine = torch.tensor([rank, cap, seed, rb.capacity], dtype=torch.int64, device=dev)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine, group=pg)
        shards = [dict(rank=int(v[0]), capacity=int(v[1]), seed=int(v[2])) for v in everyone]
        # what the communicator itself says it spans (RCCL at N > 1; gloo in a dry run)
        allreduce["communicator"] = {"backend": dist.get_backend(pg), "ranks": dist.get_world_size(pg),
                                     "rccl_version": (".".join(map(str, torch.cuda.nccl.version()))
                                                      if not job.dry else None)}
    else:
        shards = [dict(rank=0, capacity=cap, seed=seed)]

    out = None
    if rank == 0:
        kflops = sum(p[2] for p in ev_pairs)
        kbytes = sum(p[3] for p in ev_pairs)
        kms = sum(p[0].elapsed_time(p[1]) for p in ev_pairs)
        achieved = kflops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        updates_per_s = world * steps / dt
        per_update = sum(flops_per_update(cfg, first_step + i) for i in range(steps)) / steps
        # the dominant kernel: all stride-1 forward layers of two minibatches in one launch when the batch sizes allow,
        # else one launch per layer
        stacked = B % job.cu_count() == 0  # (ops.stack_granule(): one workgroup per CU owns its samples)
        # which form the stride-1 forward takes (option s1_fwd, curla_amd/csrc/conv.hip launch_rw_fwd): auto = b3, the
        # bf16-matrix-core form with fp32 operands split into three bf16 parts (conv_rwb.h); f23 / f43 = Winograd on the
        # f32-input MFMA
        from curla_amd import _lib as _clib
        s1 = "auto" if job.dry else _clib.get_option("s1_fwd")
        kname, wino, wino_factor, mfma, peak_tf, terms = {
            "auto": ("conv_rwb_fwd_kernel", "F(2,3)", 1.5, "bf16 MFMA 16x16x32, fp32 operands as three bf16 parts", PEAK_BF16_TFLOPS, 6),
            "b3": ("conv_rwb_fwd_kernel", "F(2,3)", 1.5, "bf16 MFMA 16x16x32, fp32 operands as three bf16 parts", PEAK_BF16_TFLOPS, 6),
            "f23": ("conv_rw_fwd_kernel", "F(2,3)", 1.5, "f32 MFMA 16x16x4", PEAK_F32_TFLOPS, 1),
            "f43": ("conv_rw43_fwd_kernel", "F(4,3)", 2.0, "f32 MFMA 16x16x4", PEAK_F32_TFLOPS, 1)}[s1]
        issued = achieved / wino_factor * terms  # what the matrix pipe executes, in its own FLOPs
        traffic, mfma_busy, pmc_note = (None, None, "dry run") if job.dry else committed_counters(name, kname)
        pmc_live = False
        if (not job.dry and not args.no_live_pmc and world == 1 and name == (args.config or "c2") and with_cpu_baseline):
            # (the headline configuration of a plain one-GPU run: counters measured live, by child runs under rocprofv3)
            torch.cuda.empty_cache()
            live = live_counters(name, kname)
            if live is not None:
                traffic, mfma_busy, pmc_note = live
                pmc_live = True
        opts = {} if job.dry else {k: _clib.get_option(k) for k in ("s1_fwd", "s1_wgrad", "conv1_u8", "wgrad1_u8")}
        pipe_s = sum(matrix_pipe_seconds_per_update(cfg, first_step + i, opts) for i in range(steps)) / steps
        clock = None if (job.dry or name != "c2") else committed_clock(kname)
        n_launch = max(1, len(ev_pairs))
        avg_ms = kms / n_launch
        out = {
            "metric": cfg["metric"],
            "value": updates_per_s,
            "unit": f"batch-{B} gradient updates/s (sum over ranks)",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * dt / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "float32 tensors and float32-accurate arithmetic throughout (parity 1e-4 against the float32 reference, "
                          "float64 arbiter in tests/test_gpu_fullsize.py); the stride-1 conv forward runs on the bf16 matrix "
                          "cores with every fp32 operand as the exact sum of three bf16 parts (six products per fp32 product, "
                          "fp32 accumulation: measured error below a float32 fmaf chain's, tools/micro/bf16x3_error.py); "
                          "everything else on the f32-input MFMA / VALU",
            "config": {"workload": cfg["workload"],
                       "baseline_config": (f"configs[{cfg['baseline_index']}]" if cfg["baseline_index"] is not None else
                                           "none (the reference's shipped geometry, beside BASELINE.json's)"),
                       "replay_capacity": cap * world, "shards": shards, "prefill": prefill,
                       "parallelism": f"dp{world}", "priming_updates": 1, "clock_warmup_s": args.clock_warmup_s,
                       "update_graphs": graphed},
            "transitions_per_s": updates_per_s * B,
            "conv_algorithmic_gflop_per_update": per_update / 1e9,
            "conv_roofline_frac_whole_update": per_update * (steps / dt) / (PEAK_F32_TFLOPS * 1e12),
            "conv_roofline_note": "SURVEY.md 8d accounting (the north star's 'fraction of the conv roofline'): direct-conv "
                                  "FLOPs of the update / time against the 157.3 TFLOP/s fp32 peak.  Read it as the SPEED-UP "
                                  "over what the fp32 pipe could do at best, not as a fraction bounded by 1: the kernels skip "
                                  "work (Winograd) and run most of it on the bf16 matrix cores.  matrix_pipe_time_frac_whole_"
                                  "update is the bounded figure",
            "matrix_pipe_time_frac_whole_update": pipe_s / (dt / steps),
            "matrix_pipe_time_note": "time the two matrix pipes would need at their dense peaks (bf16 2.5 PFLOP/s, f32-input "
                                     "157.3 TFLOP/s) for the conv work as the kernels issue it (Winograd factors and bf16x3 "
                                     "products counted, padding not) / measured time per update",
            "roofline": {"bound": "mfma",
                         "kernel": (f"{kname} (row-walk Winograd {wino} along x: all 3x3 s1 32->32 + bias + ReLU layers of "
                                    f"two minibatches per launch, {mfma})")
                                   if stacked else
                                   f"{kname} (row-walk Winograd {wino} along x: one 3x3 s1 32->32 + bias + ReLU layer per "
                                   f"launch, {mfma})",
                         # The kernel is Winograd F(2,3) / F(4,3) along x: it issues 1/1.5 resp. 1/2 of the direct-
                         # convolution FLOPs of SURVEY.md 8(d).  `achieved` / `frac` are what the matrix pipe really
                         # executes (compare with 1.0 and with `mfma_busy_frac_pmc`); the SURVEY 8(d) accounting in
                         # direct-conv FLOPs, which can exceed 1 because the algorithm skips work, is under `direct_equiv_*`.
                         "achieved": issued, "peak": peak_tf, "unit": "TFLOP/s",
                         "frac": issued / peak_tf, "traffic": traffic,
                         "achieved_note": (f"MFMA FLOPs issued = algorithmic direct-conv FLOPs / {wino_factor} (Winograd "
                                           f"{wino} in one dimension; strip padding not counted)"
                                           + (f" x {terms} (six bf16 x bf16 products per fp32 product: x = xh + xm + xl exactly, the "
                                              "three products below 2^-24 dropped), against the dense bf16 MFMA peak" if terms > 1 else "")),
                         "direct_equiv_achieved": achieved, "direct_equiv_frac": achieved / PEAK_F32_TFLOPS,
                         "direct_equiv_note": "direct-conv TFLOP/s of the launch and their ratio to the fp32 peak: a speed-up "
                                              "over that pipe's best case, not a utilisation (see frac)",
                         "in_kernel_clock": clock,
                         "traffic_unit": "HBM bytes per launch; " + pmc_note, "pmc_measured_live": pmc_live,
                         "algorithmic_bytes_per_launch": kbytes / n_launch,
                         "launches": len(ev_pairs), "avg_launch_ms": avg_ms,
                         "launch_timing": ("HIP events around the kernel's launches in 8 eager updates right after the "
                                           "timed region (the timed updates are graph replays)" if graphed else
                                           "HIP events around the kernel's launches inside the timed region"),
                         "hbm_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None,
                         # the kernel's other roof: its ALGORITHMIC bytes (every activation read and written once) / its
                         # launch time against the 8 TB/s HBM peak.  Round 6 found the time per sample the same at 1.7 and
                         # 2.33 GHz: this, not the matrix pipe, is what bounds it (torch's streaming copy / add kernels
                         # reach 4.8 / 6.0 TB/s on this chip: tools/hbm_mixed.py)
                         "hbm_algorithmic_GBps": (kbytes / n_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None,
                         "hbm_algorithmic_frac": (kbytes / n_launch) / (avg_ms * 1e-3) / 8e12 if avg_ms > 0 else None,
                         "hbm_peak_GBps": 8000.0,
                         "mfma_busy_frac_pmc": mfma_busy},
        }
        if job.dry:
            out["dry_run"] = True
            out["kernel_calls_traced"] = job.launches - launches0
        if allreduce is not None:
            out["allreduce"] = allreduce
    # release this configuration's HBM before the next one (or the CPU baseline) starts
    del agent, rb, L
    if not job.dry:
        torch.cuda.empty_cache()
    if out is not None and with_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, name)
    return out


_JSON_FD = None


def own_stdout():
    """stdout carries the JSON line(s) and nothing else: from here on everything any library writes to file descriptor
    1 -- gloo's and RCCL's C++ side report their connections there, several ranks at once, fragments and all -- goes to
    stderr, and emit() writes to the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="measure this configuration only (default: c2, then c3, c5 and c1t under other_configs)")
    ap.add_argument("--no-others", action="store_true", help="without --config: measure c2 only")
    ap.add_argument("--others", action="store_true",
                    help="measure c3 and c5 after c2 at N > 1 as well (default: only at N = 1 -- a multi-GPU run that "
                         "fails in a secondary configuration would take the headline line down with it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic / mfma_busy_frac_pmc live (child runs under rocprofv3 --pmc, ~1 "
                         "minute): quote the committed summaries of profiles/ instead")
    ap.add_argument("--capacity", type=int, default=CAPACITY)
    ap.add_argument("--prefill", choices=("device", "host"), default="device")
    ap.add_argument("--clock-warmup-s", type=float, default=0.6,
                    help="seconds of scratch conv launches before the warm-up steps (0 for counter-collection runs)")
    ap.add_argument("--graphs", action="store_true",
                    help="replay update() from captured hipGraphs (CurlSacAgent.enable_update_graphs; one rank, uint8-ring "
                         "configurations c2 / c3); the dominant kernel's HIP-event timing then comes from 8 eager updates "
                         "right after the timed region")
    ap.add_argument("--schedule", choices=("auto", "blocking", "overlapped"), default="auto",
                    help="data-parallel schedule at N > 1 (auto: measure both, report the faster, disclose both)")
    ap.add_argument("--sweep", action="store_true",
                    help="one job measures the sub-groups of 1, 2, 4, ... N ranks, both schedules each: a JSON line "
                         "per n and a summary line")
    ap.add_argument("--dry-run", action="store_true",
                    help="gloo + launch-trace hook on the CPU: the host path of an N-rank run without a GPU")
    args = ap.parse_args()
    main_cfg = args.config or "c2"
    world_env = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    others = [] if (args.config is not None or args.no_others or (world_env > 1 and not args.others)) else \
        ["c3", "c5", "c1t"]

    def budget(name):
        """(steps, warmup) of a configuration: the command line's for the main one and c3; c5's updates are 15x
        longer, so it is capped at 20 + 3 (about a second of GPU time)."""
        k = args.steps if args.steps is not None else (200 if name != "c5" else 20)
        w = args.warmup if args.warmup is not None else (20 if name != "c5" else 3)
        if name == "c5" and name != args.config:
            k, w = min(k, 20), min(w, 3)
        return k, w

    if "RANK" not in os.environ and args.gpus > 1:
        respawn_under_torchrun(args)
    own_stdout()
    job = Job(args)
    cpu_bl = not args.no_cpu_baseline and job.world == 1 and not job.dry and not args.sweep
    try:
        if args.sweep:
            sweep(job, main_cfg, others, budget)
            return
        out = measure_schedules(job, main_cfg, *budget(main_cfg), cpu_bl)
        extra = {}
        for name in others:
            r = measure_schedules(job, name, *budget(name), cpu_bl)
            if r is not None:
                extra[name] = r
        if job.rank == 0:
            if extra:
                out["other_configs"] = extra
            emit(out)
    finally:
        job.close()


def measure_schedules(job, name, steps, warmup, with_cpu_baseline, pg=None, n=None, single_gpu_ms=None):
    """measure() under the data-parallel schedule(s) ``--schedule`` asks for.  One rank (or a world of one): there is no
    schedule to choose.  ``auto`` measures blocking then overlapped -- each its own agent, ring and K timed steps -- and
    returns the faster one's line with both under ``allreduce``; an error in the second one is reported, not raised
    (the first result stands).  ``single_gpu_ms``: this job's own N = 1 time per step, when it has one (the sweep), for
    the communication each schedule leaves exposed."""
    world = job.world if n is None else n
    if not (job.distributed and (world > 1 or os.environ.get("CURLA_BENCH_FORCE_DIST") == "1")):
        return measure(job, name, steps, warmup, with_cpu_baseline, pg, n)
    want = job.args.schedule
    scheds = ["blocking", "overlapped"] if want == "auto" else [want]
    res = {}
    for i, sc in enumerate(scheds):
        try:
            res[sc] = measure(job, name, steps, warmup, with_cpu_baseline and i == 0, pg, n, overlap=(sc == "overlapped"))
        except Exception as e:  # noqa: BLE001
            if i == 0:
                raise
            res[sc] = {"error": f"{type(e).__name__}: {e}"}
    if job.rank != 0:
        return None
    ok = {sc: r for sc, r in res.items() if r is not None and "error" not in r}
    best = max(ok, key=lambda sc: ok[sc]["value"])
    out = ok[best]
    raw = out.pop("allreduce")

    def brief(r):
        if r is None:
            return None
        if "error" in r:
            return r
        b = {"value": r["value"], "ms_per_step": r["ms_per_step"]}
        if single_gpu_ms is not None:
            b["exposed_comm_ms_per_step"] = r["ms_per_step"] - single_gpu_ms
        return b
    out["schedule"] = best
    out["allreduce"] = {"schedule": best, "schedule_chosen_by": ("measurement (--schedule auto: both timed, the faster "
                                                                 "one's K steps are this line's value)"
                                                                 if want == "auto" else f"--schedule {want}"),
                        "blocking": brief(res.get("blocking")), "overlapped": brief(res.get("overlapped")),
                        "buckets": raw["buckets"], "communicator": raw["communicator"]}
    return out


def sweep(job, main_cfg, others, budget):
    """--sweep: sub-groups of 1, 2, 4, ... ranks of ONE job, both schedules each; one JSON line per n, then a summary."""
    import torch.distributed as dist
    world = job.world
    sizes = [n for n in (1, 2, 4, 8, 16, 32, 64) if n < world] + [world]
    groups = {}
    if job.distributed:
        # every rank creates every group (members or not), in the same order
        job.ctl = dist.new_group(backend="gloo")
        for n in sizes:
            if 1 < n < world:
                groups[n] = dist.new_group(ranks=list(range(n)))
    lines = {}
    for name in [main_cfg] + list(others):
        single_ms = None
        for n in sizes:
            r = None
            if job.rank < n:
                r = measure_schedules(job, name, *budget(name), False, groups.get(n), n, single_ms)
            job.host_barrier()
            if job.distributed:  # every rank learns the N = 1 time (rank 0 has it)
                t = torch.tensor([r["ms_per_step"] if (job.rank == 0 and n == 1) else 0.0], dtype=torch.float64)
                if n == 1:
                    dist.broadcast(t, src=0, group=job.ctl)
                    single_ms = float(t.item())
            elif n == 1:
                single_ms = r["ms_per_step"]
            if job.rank == 0:
                r["sweep"] = {"job_ranks": world, "ranks_measuring": list(range(n)),
                              "single_gpu_ms_per_step": single_ms}
                if n > 1:
                    one = lines[(name, 1)]
                    r["sweep"]["scaling_efficiency_vs_this_jobs_n1"] = r["value"] / (n * one["value"])
                lines[(name, n)] = r
                emit(r)
    if job.rank == 0:
        summary = {"sweep_summary": True, "job_ranks": world, "dry_run": bool(job.dry), "configs": {}}
        for name in [main_cfg] + list(others):
            per = {}
            for n in sizes:
                r = lines[(name, n)]
                per[str(n)] = {"value": r["value"], "ms_per_step": r["ms_per_step"], "schedule": r.get("schedule"),
                               "scaling_efficiency": r["value"] / (n * lines[(name, 1)]["value"]),
                               "allreduce": r.get("allreduce")}
            chosen = lines[(name, sizes[-1])].get("schedule")
            summary["configs"][name] = {
                "metric": lines[(name, 1)]["metric"], "unit": lines[(name, 1)]["unit"], "per_n": per,
                "default_schedule": chosen,
                "how_to_apply": (None if chosen is None else
                                 ("CURLA_DP_OVERLAP=1 / enable_data_parallel(overlap=True)" if chosen == "overlapped"
                                  else "the default (CURLA_DP_OVERLAP unset / overlap=False)"))}
        emit(summary)


if __name__ == "__main__":
    main()
