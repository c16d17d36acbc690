"""bench.py's N>1 host path, rehearsed without a GPU (``--dry-run``: gloo, kernel calls on the launch-trace hook).

The driver starts the multi-GPU bench as ``python -m torch.distributed.run ... bench.py --gpus N``; a bare
``python bench.py --gpus N`` respawns itself that way.  What must hold at any N is checked here at N = 4 (all three
configurations, through the self-respawn) and N = 8 (configs[1] only, through the driver's own launcher command):
N ranks came up, the ring is sharded ``capacity // N`` per rank, every rank samples from its own seed, exactly one
JSON line is printed, it says ``n_gpus: N`` / ``dpN`` and carries an all-reduce entry per gradient bucket."""
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


def _run(cmd):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    # gloo's C++ side reports its connections on stdout ("[Gloo] Rank r is connected to ..."); anything else on
    # stdout must be the ONE JSON line, from rank 0 only
    # (the ranks' reports interleave, so they are recognised by their text, not by their first characters)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip() and "peer ranks" not in ln and "[Gloo]" not in ln]
    # (a library's report can also land on the same line as another rank's: what must hold is ONE JSON object, and
    # nothing on stdout that is not a Gloo connection report)
    objs = [ln for ln in lines if ln.lstrip().startswith("{")]
    assert len(objs) == 1 and len(lines) == 1, lines
    return json.loads(objs[0])


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
    assert set(ar) == {"overlapped_with_backward"} | set(buckets)
    for b in buckets:
        assert ar[b]["bytes"] > 0 and ar[b]["ms"] > 0 and ar[b]["bus_GBps"] > 0
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
        assert d["allreduce"][bucket]["bytes"] // 4 in range(floats, floats + 96), (bucket, d["allreduce"][bucket])
    others = d["other_configs"]
    assert sorted(others) == ["c3", "c5"]
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


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--dry-run"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
