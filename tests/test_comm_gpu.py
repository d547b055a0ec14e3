"""The RCCL pose all-gather behind the C ABI (include/kfpos.h: kfpos_comm_*, kfpos_allgather_poses*) on the one card a
GPU box has: a communicator of world size 1 (ncclCommInitRank and ncclCommInitAll forms: init, gather = pack +
ncclAllGather + assemble, wait / sync, destroy), the assembly kernel on blocks gathered elsewhere with unequal shards,
and dist.PoseGather's "cabi" engine inside ShardedReplay against its torch engine. RCCL refuses two ranks on one device,
so world > 1 cannot run here; what differs at world > 1 is inside ncclAllGather."""
import ctypes as C

import numpy as np
import pytest

from conftest import has_gpu
from roskfpos_amd.synth import Workload

pytestmark = pytest.mark.gpu


def _need():
    if not has_gpu():
        pytest.skip("no GPU")
    import torch
    from roskfpos_amd import capi
    return torch, capi


def test_world_one_communicator_gathers_handle_positions_and_trajectories():
    torch, capi = _need()
    T, A, S = 3001, 8, 6
    w = Workload(T, A)
    bank = capi.KfposBank(capi.MODEL_TOA, T, w.anchors, init_pos=w.init_positions())
    comm = capi.KfposComm(1, 0, capi.comm_unique_id(), device=0)
    assert (comm.world, comm.rank) == (1, 0)
    dev = "cuda:0"
    out = torch.zeros(3, T, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    with pytest.raises(capi.KfposError, match="kfpos_comm_set_total"):
        comm.allgather(out, bank=bank, stream=stream)
    with pytest.raises(capi.KfposError):
        comm.set_total(0)
    assert comm.set_total(T) == (0, T)
    from roskfpos_amd.dist import device_trace
    tr = device_trace(torch, w, S, dev, False, np.float64)
    bank.run_trace_dev(S, tr["ranges"], A * T, tr["err"], 0, tr["dts"], trajectory=tr["traj"], stream=stream)
    # (1) the handle's current positions (pos_local = NULL)
    comm.allgather(out, bank=bank, stream=stream)
    comm.wait(stream)
    torch.cuda.synchronize()
    x, _, _ = bank.get_state()
    assert np.array_equal(out.cpu().numpy().T, x[:, :3])
    # (2) a whole trajectory block, three gathers in flight before anything waits (the third waits for the first) -- with
    # RCCL's all-gather and with the direct exchange (at world 1: this rank's own block, which never travels)
    assert comm.algorithm == capi.GATHER_COLLECTIVE
    for algo in (capi.GATHER_COLLECTIVE, capi.GATHER_DIRECT, capi.GATHER_COLLECTIVE):
        comm.set_algorithm(algo)
        assert comm.algorithm == algo
        outs = [torch.zeros(3 * S, T, dtype=torch.float64, device=dev) for _ in range(3)]
        for o in outs:
            comm.allgather(o, pos_local=tr["traj"], rows=3 * S, stream=stream)
        comm.sync()
        for o in outs:
            assert torch.equal(o.view(S, 3, T), tr["traj"])
    with pytest.raises(capi.KfposError):
        comm.set_algorithm(7)
    # (3) a handle that is not this communicator's shard is refused
    other = capi.KfposBank(capi.MODEL_TOA, T - 1, w.anchors, init_pos=np.zeros(3))
    with pytest.raises(capi.KfposError, match="disagree"):
        comm.allgather(out, bank=other, stream=stream)
    other.close()
    comm.close()
    bank.close()


def test_single_process_form_comm_create_all_and_grouped_gather():
    torch, capi = _need()
    T = 777
    comms = capi.KfposComm.create_all([0])
    assert len(comms) == 1 and comms[0].world == 1
    comms[0].set_total(T)
    src = torch.arange(6 * T, dtype=torch.float64, device="cuda:0").reshape(6, T)
    out = torch.zeros(6, T, dtype=torch.float64, device="cuda:0")
    capi.allgather_poses_multi(comms, [out], pos_local=[src], rows=6, streams=[torch.cuda.current_stream().cuda_stream])
    comms[0].sync()
    assert torch.equal(out, src)
    comms[0].close()


@pytest.mark.parametrize("world,total,rows", [(3, 1000, 3), (8, 1048576 + 5, 3), (4, 4096, 12), (7, 50, 6)])
def test_assembly_kernel_on_unequal_shards(world, total, rows):
    """staged [world][rows][t_pad] as any all-gather leaves it -> [rows][total]: the padding column of the smaller
    shards is dropped, global tag order restored."""
    torch, capi = _need()
    from roskfpos_amd.dist import shard_range, shard_sizes
    sizes = shard_sizes(total, world)
    t_pad = max(sizes)
    g = torch.Generator(device="cpu").manual_seed(world * 1000 + rows)
    truth = torch.rand(rows, total, dtype=torch.float64, generator=g)
    staged = torch.full((world, rows, t_pad), -7.0, dtype=torch.float64)
    for r in range(world):
        lo, hi = shard_range(total, world, r)
        staged[r, :, :hi - lo] = truth[:, lo:hi]
    out = torch.zeros(rows, total, dtype=torch.float64, device="cuda:0")
    capi.assemble_poses_dev(world, rows, total, staged.to("cuda:0"), out, device=0,
                            stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), truth)


@pytest.mark.parametrize("mode,epl", [("epoch", 1), ("launch", 5), ("trajectory", 4)])
def test_pose_gather_cabi_engine_equals_torch_engine(mode, epl):
    torch, capi = _need()
    from roskfpos_amd.dist import ShardedReplay, device_trace
    T, A, S = 5000, 8, 10
    w = Workload(T, A)
    dev = "cuda:0"
    stream = torch.cuda.current_stream().cuda_stream
    got = {}
    for engine in ("torch", "cabi"):
        tr = device_trace(torch, w, S, dev, False, np.float64)
        bank = capi.KfposBank(capi.MODEL_TOA, T, w.anchors, init_pos=w.init_positions())
        rep = ShardedReplay(bank, T, dev, gather_mode=mode, epochs_per_launch=epl, stream=stream, engine=engine)
        assert rep.gather.engine == engine
        blocks = {}

        def keep(first, cnt, full, rep=rep, blocks=blocks):
            whole = rep.gather.assemble(full).cpu().numpy()
            for k in range(cnt):
                blocks[first + k] = whole[3 * k:3 * k + 3].copy()

        rep.run(tr, 0, S, on_gathered=keep)
        assert rep.gather.self_check()      # the predictable-pattern gather bench.py runs before trusting an engine
        got[engine] = blocks
        rep.gather.close()
        bank.close()
    assert sorted(got["torch"]) == sorted(got["cabi"]) and len(got["cabi"]) >= S // epl
    for s in got["torch"]:
        assert np.array_equal(got["torch"][s], got["cabi"][s])


def test_cpp_node_shards_without_python(tmp_path):
    """tools/shard_node.cpp: a C++ multi-tag node on the C ABI alone -- kfpos_comm_create_all, one handle per device,
    kfpos_allgather_poses_multi once per epoch -- built with hipcc and run on the one device of this box. Its exit status
    says whether every device received exactly the poses the shards' own getPose reports."""
    if not has_gpu():
        pytest.skip("no GPU")
    import json
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "roskfpos_amd", "csrc")
    exe = str(tmp_path / "shard_node")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-I", os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "tools", "shard_node.cpp"), "-L", csrc, "-lkfpos_hip",
                           "-Wl,-rpath," + csrc])
    res = subprocess.run([exe, "1", "50001", "12"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert d["devices"] == 1 and d["mismatches"] == 0 and d["rccl_version"] >= 20000


def test_algorithm_calibration_runs_both_exchanges_and_picks_one():
    """dist._pick_algorithm -- what bench.py runs before its timed region when every rank has a GPU -- in a process group
    of one: both ways of moving the blocks are checked with the predictable pattern, timed, and one stays selected."""
    if not has_gpu():
        pytest.skip("no GPU")
    import os
    import socket
    import torch
    import torch.distributed as dist
    from roskfpos_amd import capi
    from roskfpos_amd.dist import PoseGather, _pick_algorithm
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = PoseGather(4099, "cuda:0", rows=6, engine="cabi")
        cal = _pick_algorithm(g, "cuda:0")
        assert cal["collective"]["ok"] and cal["direct"]["ok"] and cal["picked"] in ("collective", "direct")
        assert cal["collective"]["us_per_gather"] > 0 and cal["direct"]["us_per_gather"] > 0
        assert g.comm.algorithm == (capi.GATHER_DIRECT if cal["picked"] == "direct" else capi.GATHER_COLLECTIVE)
        assert g.self_check()
        g.close()
    finally:
        dist.destroy_process_group()
