"""bench.py started the way the driver starts it -- plainly, no launcher -- with real banks on the GPU box.
N = 2 on a one-GPU box: bench.py spawns its two ranks itself (before it touches the GPU), the ranks share the card and the
pose gather goes through gloo, and the line says that this is a rehearsal of the N > 1 path, not a scaling figure."""
import json
import os
import subprocess
import sys

import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout=600, **extra_env):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(extra_env)
    res = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_plain_two_rank_command_on_one_card():
    if not has_gpu():
        pytest.skip("no GPU")
    import torch
    d = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--tags-per-gpu", "8192", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0 and d["state_finite"] and d["trajectory_matches_state"]
    assert d["config"]["total_tags"] == 16384 and d["config"]["launched_by"].startswith("bench.py itself")
    if torch.cuda.device_count() < 2:
        assert d["config"]["backend"] == "gloo" and "rehearsal" in d["config"]["note"]
    assert d["roofline"]["frac"] > 0 and d["repeats"]["kernel_us_per_launch_spread"]["min"] > 0


def test_plain_single_gpu_command_reduced_size():
    """the N = 1 line at a reduced batch (the full-size run is the driver's): every contract key, roofline and
    cpu_baseline objects present, the GPU within 1e-6 m of the oracle on the sampled tags"""
    if not has_gpu():
        pytest.skip("no GPU")
    d = _run(["--gpus", "1", "--steps", "10", "--warmup", "5", "--tags-per-gpu", "4096", "--repeats", "2"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["config"]["launched_by"] == "direct"
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == cb["host_cores"] >= 1 and cb["value"] > 0
    assert d["rms_pos_err_vs_cpu_ref_m"] <= 1e-6
    # configs[2] as BASELINE.json words it (compact covariance storage) rides along in the same line
    p48 = d["secondary"]["c3_p48"]
    assert "KFPOS_STORE_MIXED" in d["dtype"] and "double,float" in d["roofline"]["kernel"]
    assert p48["value"] > 0 and p48["rms_pos_err_vs_cpu_ref_m"] <= 1e-6 and "p48" in p48["kernel"]


def test_p48_as_the_headline_storage():
    if not has_gpu():
        pytest.skip("no GPU")
    d = _run(["--gpus", "1", "--steps", "10", "--warmup", "5", "--tags-per-gpu", "4096", "--repeats", "1"],
             KFPOS_BENCH_STORAGE="p48")
    assert "KFPOS_STORE_P48" in d["dtype"] and "p48" in d["roofline"]["kernel"] and d["rms_pos_err_vs_cpu_ref_m"] <= 1e-6
    assert "c3_p48" not in d.get("secondary", {})


def test_two_rank_command_with_the_c_abi_gather():
    """the branch an 8-GPU node takes -- make_pose_gather's "cabi" engine: communicator, self-check, calibration of the two
    exchange algorithms, kfpos_allgather_poses per launch -- driven by bench.py itself at two ranks. On a one-GPU box the
    transport underneath is the tests' stand-in for librccl (tests/fake_rccl; RCCL refuses two ranks on one device)."""
    if not has_gpu():
        pytest.skip("no GPU")
    import torch
    extra = {}
    if torch.cuda.device_count() < 2:
        from test_comm_world_gpu import FAKE_VERSION, _fake
        extra = dict(KFPOS_RCCL_PATH=_fake(), KFPOS_GATHER_ENGINE="cabi")
    d = _run(["--gpus", "2", "--config", "c4", "--total-tags", "120001", "--steps", "20", "--warmup", "5",
              "--gather", "epoch", "--no-cpu-baseline"], **extra)
    c = d["config"]
    assert d["n_gpus"] == 2 and c["total_tags"] == 120001 and d["state_finite"] and d["trajectory_matches_state"]
    assert "pose_gather_fallback" not in c, c.get("pose_gather_fallback")
    assert "kfpos_allgather_poses" in c["pose_gather"] and "torch.distributed" not in c["pose_gather"]
    cal = c["pose_gather_algorithms"]
    assert cal and all(v["collective"]["ok"] and v["direct"]["ok"] for v in cal.values())
    if extra:
        assert c["rccl_version"] == FAKE_VERSION
