"""N>1 path on CPU: tag sharding + per-epoch pose all-gather, world_size 2 (and 3), gloo backend.

The compute on each rank is the host emulation of the kernel body (no GPU here); what is under test is
roskfpos_amd.dist -- shard ranges, per-shard regeneration of the synthetic inputs, the double-buffered
gather -- and the shard-equivalence property of SURVEY.md 8e: N shards == 1 shard, bit for bit.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cases import Case
from impls import EmuImpl
from roskfpos_amd.dist import PoseGather, make_pose_gather, shard_range, shard_sizes
from roskfpos_amd.synth import Workload

T_TOTAL, A, S = 97, 8, 12  # 97: the two shards differ by one tag (49 + 48), the gather pads and trims


def test_shard_ranges_cover_the_batch():
    for total, world in [(96, 2), (65536 * 8, 8), (10, 3), (5, 8)]:
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _run_shard(tag0, n, steps):
    case = Case("shard", 0, A, T=n, S=steps)
    w = Workload(n, A, tag0=tag0)
    f = EmuImpl(case, w, w.init_positions())
    out = []
    for s in range(steps):
        f.step_toa(w.ranges_mm(s), w.err_est(), w.dt_of(s))
        out.append(f.pose(0.0)[0].copy())
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(T_TOTAL, world, rank)
    poses = _run_shard(lo, hi - lo, S)
    g = PoseGather(hi - lo, "cpu", sizes=shard_sizes(T_TOTAL, world))
    assert g.uniform == (T_TOTAL % world == 0) and g.t_pad == -(-T_TOTAL // world)
    gathered = []
    for s in range(S):
        buf = g.buffer()
        buf[:, :hi - lo].copy_(torch.from_numpy(np.ascontiguousarray(poses[s].T)))  # component-major [3][T_local]
        full = g.gather()
        g.wait()
        gathered.append(g.assemble(full).clone().numpy())   # [3][T_TOTAL], padding dropped
    # one collective carrying a whole block of epochs ("trajectory" mode of ShardedReplay)
    gt = PoseGather(hi - lo, "cpu", rows=3 * S, sizes=shard_sizes(T_TOTAL, world))
    gt.buffer()[:, :hi - lo].copy_(torch.from_numpy(np.concatenate([p.T for p in poses])))
    block = gt.assemble(gt.gather()).clone().numpy()          # [3 S][T_TOTAL]
    gt.wait()
    # the safety net of the C-ABI engine: asked for where it cannot work (CPU tensors), every rank must agree to fall
    # back to torch.distributed's collective, say why, and still gather correctly
    assert g.self_check() and gt.self_check()
    gf = make_pose_gather(hi - lo, "cpu", rows=3, sizes=shard_sizes(T_TOTAL, world), engine="cabi")
    assert gf.engine == "torch" and "HBM" in gf.fallback_reason and gf.self_check()
    if rank == 0:
        q.put((np.stack(gathered), block))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_of_unequal_shards_equals_single_shard(world):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, block = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = np.stack(_run_shard(0, T_TOTAL, S))                      # [S][T][3]
    assert got.shape == (S, 3, T_TOTAL)
    assert np.array_equal(got.transpose(0, 2, 1), single)              # bit-for-bit
    assert np.array_equal(block.reshape(S, 3, T_TOTAL), got)


def test_pose_gather_rejects_wrong_sizes():
    with pytest.raises(ValueError):
        PoseGather(10, "cpu", sizes=[11])
