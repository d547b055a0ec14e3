"""The pose exchange of the C ABI (include/kfpos.h: kfpos_comm_*, kfpos_allgather_poses*) at world sizes 2, 3 and 4 on the
one card of a GPU box. RCCL refuses two ranks on one device, so the transport here is the tests' stand-in
(tests/fake_rccl: the dozen entry points kfpos_comm.hip resolves, over host shared memory, strict about what the ranks
post), selected with KFPOS_RCCL_PATH; everything in front of and behind the transport is the product: the packing of
unequal shards, the peer loop of the direct exchange, the grouped single-process form, the assembly kernel, the event
order of the two buffer sets, dist.PoseGather's "cabi" engine with its self-check and calibration, ShardedReplay.
What stays unexercised is RCCL's own machinery at world > 1.

librccl is resolved once per process, so every case runs in child processes that see KFPOS_RCCL_PATH from the start.
"""
import ctypes as C
import json
import os
import shutil
import socket
import subprocess

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
FAKE = os.path.join(FAKE_DIR, "libfake_rccl.so")
FAKE_VERSION = 99999


def _fake():
    if not has_gpu():
        pytest.skip("no GPU")
    src = os.path.join(FAKE_DIR, "fake_rccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        subprocess.check_call([shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC",
                               "-o", FAKE, src, "-lrt"])
    return FAKE


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_children(target, world, args, timeout=600):
    import torch.multiprocessing as mp
    os.environ["KFPOS_RCCL_PATH"] = _fake()
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "60"
    try:
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=target, args=(r, world) + tuple(args)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=timeout)
            if p.is_alive():
                p.terminate()
                p.join(10)
            assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    finally:
        os.environ.pop("KFPOS_RCCL_PATH", None)
        os.environ.pop("FAKE_RCCL_TIMEOUT_S", None)


def _stats():
    out = (C.c_uint64 * 4)()
    C.CDLL(FAKE).fake_rccl_stats(out)
    return dict(rounds=out[0], bytes=out[1], allgathers=out[2], p2p=out[3])


# ---- one process, several shards: kfpos_comm_create_all + kfpos_allgather_poses_multi ----
def _one_process(_rank, _one, world, outfile):
    import torch
    from roskfpos_amd import capi
    assert capi.load().kfpos_comm_backend_version() == FAKE_VERSION
    dev = "cuda:0"
    stream = torch.cuda.current_stream().cuda_stream
    total = world * 5000 + (world - 1)                   # all ranks but the last hold one tag more
    comms = capi.KfposComm.create_all([0] * world)
    spans = [c.set_total(total) for c in comms]
    assert spans == [capi.shard_range(total, world, r) for r in range(world)]
    seen = []
    for algo in (capi.GATHER_COLLECTIVE, capi.GATHER_DIRECT, capi.GATHER_COLLECTIVE):
        for c in comms:
            c.set_algorithm(algo)
        for rows in (3, 12, 3):
            before = _stats()
            calls = []
            for k in range(3):                            # three in flight: the third reuses the first buffer set
                whole = (torch.arange(rows * total, dtype=torch.float64, device=dev).reshape(rows, total)
                         * (k + 1) + 0.5 * rows)
                src = [whole[:, lo:hi].contiguous() for lo, hi in spans]
                outs = [torch.full((rows, total), -1.0, dtype=torch.float64, device=dev) for _ in range(world)]
                capi.allgather_poses_multi(comms, outs, pos_local=src, rows=rows, streams=[stream] * world)
                for s in src:
                    s.fill_(float("nan"))                 # the source may be overwritten once the call returned
                calls.append((whole, outs))
            for c in comms:
                c.sync()
            for whole, outs in calls:
                for o in outs:
                    assert torch.equal(o, whole)
            after = _stats()
            d = {k: after[k] - before[k] for k in after}
            assert d["rounds"] == 3
            if algo == capi.GATHER_DIRECT:                # every rank sends to and receives from every other one
                assert d["allgathers"] == 0 and d["p2p"] == 3 * 2 * world * (world - 1)
                assert d["bytes"] == 3 * world * (world - 1) * rows * (total // world + 1) * 8
            else:
                assert d["p2p"] == 0 and d["allgathers"] == 3 * world
            seen.append(d)
    # a handle's own positions as the source (pos_local = NULL), one bank per shard
    from roskfpos_amd.synth import Workload
    banks, want = [], np.zeros((3, total))
    for lo, hi in spans:
        w = Workload(hi - lo, 8, tag0=lo)
        b = capi.KfposBank(capi.MODEL_TOA, hi - lo, w.anchors, init_pos=w.init_positions())
        b.step_toa(w.ranges_mm(0), w.err_est(), 0.1)
        want[:, lo:hi] = b.get_state()[0][:, :3].T
        banks.append(b)
    outs = [torch.zeros(3, total, dtype=torch.float64, device=dev) for _ in range(world)]
    capi.allgather_poses_multi(comms, outs, banks=banks, streams=[stream] * world)
    for c in comms:
        c.sync()
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), want)
    for b in banks:
        b.close()
    for c in comms:
        c.close()
    json.dump(seen, open(outfile, "w"))


@pytest.mark.parametrize("world", [2, 4])
def test_grouped_exchange_of_several_shards_in_one_process(world, tmp_path):
    out = str(tmp_path / "stats.json")
    _run_children(_one_process, 1, (world, out))
    assert len(json.load(open(out))) == 9


# ---- one process per rank: kfpos_comm_create + kfpos_allgather_poses through dist.PoseGather / ShardedReplay ----
def _per_rank(rank, world, port, total, steps, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    from roskfpos_amd import capi
    from roskfpos_amd.dist import ShardedReplay, device_trace, make_pose_gather, shard_range, shard_sizes
    from roskfpos_amd.synth import Workload
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev, A = "cuda:0", 8
    lo, hi = shard_range(total, world, rank)
    w = Workload(hi - lo, A, tag0=lo)
    trace = device_trace(torch, w, steps, dev, False, np.float64)
    stream = torch.cuda.current_stream().cuda_stream
    got, meta = {}, {}
    for mode, epl in (("epoch", 1), ("trajectory", 4), ("launch", 5)):
        rows = 3 * epl if mode == "trajectory" else 3
        g = make_pose_gather(hi - lo, dev, rows=rows, sizes=shard_sizes(total, world), engine="cabi")
        assert g.engine == "cabi" and g.fallback_reason is None, g.fallback_reason
        cal = g.calibration
        assert cal["collective"]["ok"] and cal["direct"]["ok"] and cal["picked"] in ("collective", "direct")
        meta[mode] = cal
        for algo in (capi.GATHER_COLLECTIVE, capi.GATHER_DIRECT):
            g.comm.set_algorithm(algo)
            bank = capi.KfposBank(capi.MODEL_TOA, hi - lo, w.anchors, storage=capi.STORE_F64, init_pos=w.init_positions())
            rep = ShardedReplay(bank, total, dev, gather_mode=mode, epochs_per_launch=epl, stream=stream, gather=g)
            blocks = {}

            def keep(first, cnt, full, blocks=blocks):
                whole = full.cpu().numpy()               # engine "cabi": already [rows][total]
                for k in range(cnt):
                    blocks[first + k] = whole[3 * k:3 * k + 3].copy()

            rep.run(trace, 0, steps, on_gathered=keep)
            got[(mode, algo)] = blocks
            bank.close()
        g.close()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"),
             **{f"{m}_{a}_{s}": v for (m, a), blocks in got.items() for s, v in blocks.items()})
    if rank == 0:
        json.dump(meta, open(os.path.join(outdir, "calibration.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def _unsharded(total, steps):
    import torch
    from roskfpos_amd import capi
    from roskfpos_amd.dist import device_trace
    from roskfpos_amd.synth import Workload
    w = Workload(total, 8)
    trace = device_trace(torch, w, steps, "cuda:0", False, np.float64)
    bank = capi.KfposBank(capi.MODEL_TOA, total, w.anchors, storage=capi.STORE_F64, init_pos=w.init_positions())
    stream = torch.cuda.current_stream().cuda_stream
    for s in range(steps):
        bank.run_trace_dev(1, trace["ranges"][s], 8 * total, trace["err"], 0, trace["dts"][s:s + 1],
                           trajectory=trace["traj"][s], stream=stream)
    torch.cuda.synchronize()
    bank.close()
    return trace["traj"].cpu().numpy()


@pytest.mark.parametrize("world,total", [(2, 2 * 30000 + 1), (3, 3 * 20000 + 2), (4, 4 * 15000 + 3)])  # one-tag-per-lane kernels on both sides
def test_ranks_in_processes_gather_what_one_bank_computes(world, total, tmp_path):
    """Every rank -- not only rank 0 -- ends up with the poses of ALL tags, bit for bit what one unsharded bank computes,
    per epoch, per launch and per trajectory block, with RCCL's all-gather and with the direct exchange."""
    steps = 10
    _fake()
    _run_children(_per_rank, world, (_free_port(), total, steps, str(tmp_path)))
    single = _unsharded(total, steps)                    # [S][3][total]
    for r in range(world):
        got = np.load(str(tmp_path / f"rank{r}.npz"))
        for algo in (0, 1):
            for s in range(steps):
                assert np.array_equal(got[f"epoch_{algo}_{s}"], single[s]), (r, algo, s)
                assert np.array_equal(got[f"trajectory_{algo}_{s}"], single[s]), (r, algo, s)
            for s in (4, 9):
                assert np.array_equal(got[f"launch_{algo}_{s}"], single[s]), (r, algo, s)
    cal = json.load(open(str(tmp_path / "calibration.json")))
    assert set(cal) == {"epoch", "trajectory", "launch"}


# ---- what the stand-in refuses: the mistakes that would hang or corrupt with the real library ----
def _mismatch(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from roskfpos_amd import capi
    dist.init_process_group("gloo", rank=rank, world_size=world)
    uid = [capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    comm = capi.KfposComm(world, rank, uid[0], device=0)
    comm.set_total(1000)
    rows = 3 if rank == 0 else 6                         # the ranks disagree about the block they exchange
    src = torch.zeros(rows, 500, dtype=torch.float64, device="cuda:0")
    out = torch.zeros(rows, 1000, dtype=torch.float64, device="cuda:0")
    try:
        comm.allgather(out, pos_local=src, rows=rows, stream=torch.cuda.current_stream().cuda_stream)
        verdict = "accepted"
    except capi.KfposError as e:
        verdict = str(e)
    open(os.path.join(outdir, f"verdict{rank}.txt"), "w").write(verdict)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_disagree_about_the_block_are_told_so(tmp_path):
    _fake()
    _run_children(_mismatch, 2, (_free_port(), str(tmp_path)))
    for r in range(2):
        v = open(str(tmp_path / f"verdict{r}.txt")).read()
        assert "communicator" in v.lower() or "rccl" in v.lower() or "invalid" in v.lower(), v
        assert v != "accepted"


# ---- the C++ node of INTEGRATION.md section 3 with four shards ----
def test_cpp_node_with_four_shards(tmp_path):
    fake = _fake()
    csrc = os.path.join(ROOT, "roskfpos_amd", "csrc")
    exe = str(tmp_path / "shard_node")
    subprocess.check_call([shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I",
                           os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tools", "shard_node.cpp"),
                           "-L", csrc, "-lkfpos_hip", "-Wl,-rpath," + csrc])
    env = dict(os.environ, KFPOS_RCCL_PATH=fake)
    for algo in ("collective", "direct"):
        res = subprocess.run([exe, "4", "40003", "8", "0,0,0,0"], capture_output=True, text=True, timeout=300,
                             env=dict(env, KFPOS_GATHER_ALGO=algo))
        assert res.returncode == 0, res.stdout + res.stderr
        d = json.loads(res.stdout.strip().splitlines()[-1])
        assert d["devices"] == 4 and d["mismatches"] == 0 and d["rccl_version"] == FAKE_VERSION
