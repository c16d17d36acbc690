"""bench.py's N>1 host path, rehearsed without a GPU (``--dry-run``: gloo, kernel calls on the launch-trace hook).

The driver starts the multi-GPU bench as ``python -m torch.distributed.run ... bench.py --gpus N``; a bare
``python bench.py --gpus N`` respawns itself that way.  What must hold at any N is checked here at N = 4 (all three
configurations, through the self-respawn) and N = 8 (configs[1] only, through the driver's own launcher command):
N ranks came up, the ring is sharded ``capacity // N`` per rank, every rank samples from its own seed, exactly one
JSON line is printed, it says ``n_gpus: N`` / ``dpN``, carries an all-reduce entry per gradient bucket and BOTH
data-parallel schedules (``--schedule auto``).  ``--sweep`` (sub-groups of 1, 2, 4, 8 ranks of one 8-rank job, both
schedules each, one line per n + a summary) is rehearsed the same way."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, n_lines=1):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    # stdout carries the JSON line(s) and NOTHING else: bench.py moves file descriptor 1 to stderr for everything but its
    # own lines (gloo's / RCCL's C++ sides report their connections on stdout, several ranks at once -- fragments of those
    # reports used to land between the lines)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    objs = [ln for ln in lines if ln.lstrip().startswith("{")]
    assert len(objs) == n_lines and len(lines) == n_lines, lines
    return json.loads(objs[0]) if n_lines == 1 else [json.loads(o) for o in objs]


def _check_line(d, n, capacity, buckets):
    assert d["dry_run"] is True and d["n_gpus"] == n and d["config"]["parallelism"] == f"dp{n}"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["steps"] >= 1 and d["value"] > 0 and abs(d["value"] - n * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) \
        <= 1e-6 * d["value"]
    shards = d["config"]["shards"]
    assert [s["rank"] for s in shards] == list(range(n))
    assert all(s["capacity"] == capacity // n for s in shards)
    assert sorted({s["seed"] for s in shards}) == [1 + r for r in range(n)]  # rank r samples from seed 1 + r
    assert d["config"]["replay_capacity"] == (capacity // n) * n
    ar = d["allreduce"]
    assert set(ar) == {"schedule", "schedule_chosen_by", "blocking", "overlapped", "buckets", "communicator"}
    assert set(ar["buckets"]) == set(buckets)
    for b in buckets:
        assert ar["buckets"][b]["bytes"] > 0 and ar["buckets"][b]["ms"] > 0 and ar["buckets"][b]["bus_GBps"] > 0
    # both schedules were timed; the line is the faster one's
    assert ar["schedule"] == d["schedule"] and ar["schedule"] in ("blocking", "overlapped")
    for sc in ("blocking", "overlapped"):
        assert ar[sc]["value"] > 0 and ar[sc]["ms_per_step"] > 0
    assert d["value"] == max(ar["blocking"]["value"], ar["overlapped"]["value"]) == ar[ar["schedule"]]["value"]
    assert ar["communicator"] == {"backend": "gloo", "ranks": n, "rccl_version": None}
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    assert d["kernel_calls_traced"] > 50 * (d["steps"] + d["warmup"] + 1) // 2


def test_four_ranks_through_the_self_respawn_all_configs():
    d = _run([sys.executable, "bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1", "--dry-run", "--capacity", "512",
              "--others"])
    _check_line(d, 4, 512, ("critic", "actor", "cpc"))
    assert d["metric"].startswith("SAC+CURL gradient updates/sec, batch=512")
    assert d["config"]["baseline_config"] == "configs[1]"
    # bucket sizes of SURVEY.md 8e at hidden 1024: critic [encoder|Q1|Q2], actor [fc, ln | trunk], cpc [W | encoder]
    # (+ the 16-byte slot padding of the flat buffers: a few floats per tensor)
    for bucket, floats in (("critic", 3_777_912), ("actor", 2_643_674), ("cpc", 1_570_618)):
        got = d["allreduce"]["buckets"][bucket]
        assert got["bytes"] // 4 in range(floats, floats + 96), (bucket, got)
    others = d["other_configs"]
    assert sorted(others) == ["c1t", "c3", "c5"]
    _check_line(others["c1t"], 4, 512, ("critic", "actor", "cpc"))
    assert "reference's shipped" in others["c1t"]["config"]["baseline_config"]
    _check_line(others["c3"], 4, 512, ("critic", "actor"))   # pixel_sac: no cpc bucket
    _check_line(others["c5"], 4, 512, ("critic", "actor", "cpc"))
    assert others["c3"]["config"]["baseline_config"] == "configs[2]"
    assert others["c5"]["config"]["baseline_config"] == "configs[4]" and "batch-1024" in others["c5"]["unit"]


def test_eight_ranks_through_the_drivers_launcher_command():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--dry-run", "--capacity", "1000"]  # (N > 1 without --others: the headline configuration only)
    d = _run(cmd)
    _check_line(d, 8, 1000, ("critic", "actor", "cpc"))
    assert "other_configs" not in d


def test_one_schedule_on_request():
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--capacity", "256",
              "--schedule", "overlapped"])
    assert d["n_gpus"] == 2 and d["schedule"] == "overlapped"
    assert d["allreduce"]["blocking"] is None and d["allreduce"]["overlapped"]["value"] == d["value"]


def test_sweep_of_an_eight_rank_job_decides_the_schedule_in_one_command():
    """``bench.py --gpus 8 --sweep``: n = 1, 2, 4, 8 sub-groups of ONE job, both schedules each, per-bucket all-reduce
    times, what the communicator says it spans, efficiency against the job's own n = 1 -- one line per n + a summary."""
    out = _run([sys.executable, "bench.py", "--gpus", "8", "--sweep", "--steps", "2", "--warmup", "1", "--dry-run",
                "--capacity", "1000"], n_lines=5)
    lines, summary = out[:4], out[4]
    assert [d["n_gpus"] for d in lines] == [1, 2, 4, 8]
    one = lines[0]
    assert "allreduce" not in one and one["sweep"]["ranks_measuring"] == [0] and one["config"]["parallelism"] == "dp1"
    assert one["config"]["shards"] == [dict(rank=0, capacity=1000, seed=1)]
    for d in lines[1:]:
        n = d["n_gpus"]
        _check_line(d, n, 1000, ("critic", "actor", "cpc"))
        sw = d["sweep"]
        assert sw["job_ranks"] == 8 and sw["ranks_measuring"] == list(range(n))
        assert sw["single_gpu_ms_per_step"] == one["ms_per_step"]
        assert abs(sw["scaling_efficiency_vs_this_jobs_n1"] - d["value"] / (n * one["value"])) < 1e-9
        for sc in ("blocking", "overlapped"):
            exposed = d["allreduce"][sc]["exposed_comm_ms_per_step"]
            assert abs(exposed - (d["allreduce"][sc]["ms_per_step"] - one["ms_per_step"])) < 1e-9
    assert summary["sweep_summary"] is True and summary["job_ranks"] == 8 and summary["dry_run"] is True
    c2 = summary["configs"]["c2"]
    assert sorted(c2["per_n"], key=int) == ["1", "2", "4", "8"]
    assert c2["per_n"]["1"]["scaling_efficiency"] == 1.0 and c2["per_n"]["1"]["schedule"] is None
    assert c2["default_schedule"] == lines[3]["schedule"] == c2["per_n"]["8"]["schedule"]
    assert ("CURLA_DP_OVERLAP=1" in c2["how_to_apply"]) == (c2["default_schedule"] == "overlapped")


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--dry-run"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
