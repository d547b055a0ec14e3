"""bench.py's N > 1 launch paths, without a GPU: `python bench.py --gpus N` started plainly has to spawn its own ranks
(the driver starts `--gpus 1` that way), and the torchrun form has to keep working. --dry-run swaps the filter bank for
a stand-in that computes nothing, so what runs here is the launcher, the environment plumbing, the process group, the
shard arithmetic, dist.ShardedReplay + PoseGather over gloo on CPU tensors, and the JSON contract. The same plain
command with real banks runs on the GPU box in test_bench_gpu.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config"}


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    return env


def _json_line(stdout: str):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry the one JSON line and nothing else: {lines!r}"
    return json.loads(lines[0])


def test_plain_command_launches_its_own_ranks():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run",
                          "--tags-per-gpu", "200", "--epochs-per-launch", "4"], capture_output=True, text=True,
                         env=_clean_env(), timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    d = _json_line(res.stdout)
    assert CONTRACT_KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["value"] is None and "dry_run" in d            # nothing was measured and the line says so
    assert d["config"]["total_tags"] == 400 and d["config"]["tags_per_gpu"] == 200
    assert d["config"]["launched_by"].startswith("bench.py itself")
    assert d["config"]["epochs_in_timed_launches"] == [4, 2]
    assert "2 collectives in the timed region" in d["config"]["pose_gather"]
    assert d["per_epoch_launch"]["pose_gathers"] == 6 and d["per_epoch_launch"]["launches"] == 6
    assert d["repeats"]["n"] == 3


def test_strong_scaling_config_unequal_shards_three_ranks():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--config", "c4", "--total-tags", "1000", "--steps", "5",
                          "--warmup", "1", "--dry-run", "--gather", "trajectory", "--repeats", "1"],
                         capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["config"]["total_tags"] == 1000
    assert d["config"]["tags_per_gpu"] == 334          # rank 0 of shard_range(1000, 3): 334 + 333 + 333


def test_torchrun_form_still_works():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "4",
                          "--warmup", "1", "--dry-run", "--tags-per-gpu", "128", "--no-per-epoch"],
                         capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["config"]["launched_by"].startswith("an outer launcher")


def test_a_failing_rank_ends_the_run_quickly():
    """Without --dry-run and without a GPU every rank refuses to start (no CPU fallback); the launcher must notice and
    return non-zero instead of waiting for a collective's timeout."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True,
                         text=True, env=_clean_env(), timeout=120)
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert "no CPU fallback" in res.stderr


def test_world_size_mismatch_is_refused():
    env = dict(_clean_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, env=env,
                         timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE=3" in res.stderr
