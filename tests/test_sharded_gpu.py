"""BASELINE configs[3] as stated -- 1 048 576 tags x 8 anchors, UWB-only 6-state filter, f64 -- sharded over 2 and 4
ranks with dist.ShardedReplay (the code bench.py --config c4 runs), all ranks on the one card of the GPU box and
exchanging poses over gloo (RCCL needs one GPU per rank; the collective is the only thing that differs).

Checked: (a) the poses gathered on rank 0 -- per epoch, per launch, and per launch with the whole trajectory block --
equal, bit for bit, what ONE unsharded bank of the same kernel family computes from the same per-tag inputs
(SURVEY.md 8e); (b) a strided sample of tags against the oracle, <= 1e-6 m RMS.
"""
import os
import socket

import numpy as np
import pytest

from conftest import has_gpu

TOTAL, A, S = 1048576, 8, 10


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, total, steps, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    from roskfpos_amd import capi
    from roskfpos_amd.dist import ShardedReplay, device_trace, shard_range
    from roskfpos_amd.synth import Workload
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = "cuda:0"
    lo, hi = shard_range(total, world, rank)
    w = Workload(hi - lo, A, tag0=lo)
    trace = device_trace(torch, w, steps, dev, False, np.float64)
    stream = torch.cuda.current_stream().cuda_stream
    got = {}
    for mode, epl in (("epoch", 1), ("trajectory", 4), ("launch", 5)):
        bank = capi.KfposBank(capi.MODEL_TOA, hi - lo, w.anchors, storage=capi.STORE_F64, init_pos=w.init_positions())
        rep = ShardedReplay(bank, total, dev, gather_mode=mode, epochs_per_launch=epl, stream=stream)
        assert rep.gather.host_staged and rep.gather.sizes == [b - a for a, b in
                                                                (shard_range(total, world, r) for r in range(world))]
        blocks = {}

        def keep(first, cnt, full, rep=rep, blocks=blocks):
            if rank == 0:
                whole = rep.gather.assemble(full).cpu().numpy()  # [rows][total]
                for k in range(cnt):
                    blocks[first + k] = whole[3 * k:3 * k + 3].copy()

        rep.run(trace, 0, steps, on_gathered=keep)
        assert rep.launches == -(-steps // epl)
        got[mode] = blocks
        bank.close()
    if rank == 0:
        np.savez(os.path.join(outdir, "gathered.npz"),
                 **{f"{m}_{s}": v for m, blocks in got.items() for s, v in blocks.items()})
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world, total, steps, outdir):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, steps, outdir)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    return np.load(os.path.join(outdir, "gathered.npz"))


def _unsharded(total, steps):
    """One bank over all tags, one launch per epoch: the trajectory [S][3][total] (host)."""
    import torch
    from roskfpos_amd import capi
    from roskfpos_amd.dist import device_trace
    from roskfpos_amd.synth import Workload
    w = Workload(total, A)
    trace = device_trace(torch, w, steps, "cuda:0", False, np.float64)
    bank = capi.KfposBank(capi.MODEL_TOA, total, w.anchors, storage=capi.STORE_F64, init_pos=w.init_positions())
    stream = torch.cuda.current_stream().cuda_stream
    for s in range(steps):
        bank.run_trace_dev(1, trace["ranges"][s], A * total, trace["err"], 0, trace["dts"][s:s + 1],
                           trajectory=trace["traj"][s], stream=stream)
    torch.cuda.synchronize()
    x, P, _ = bank.get_state()
    bank.close()
    assert np.isfinite(x).all() and np.isfinite(P).all()
    return trace["traj"].cpu().numpy()


def _oracle_sample(total, steps, n_blocks=32, block=8):
    """Oracle trajectories of n_blocks x block tags spread over the batch: (tag indices, [S][3][n])."""
    import oracle_py
    from roskfpos_amd.synth import Workload
    idx, ws = [], []
    for b in range(n_blocks):
        t0 = (b * (total // n_blocks) + 17 * b) % (total - block)
        ws.append(Workload(block, A, tag0=t0))
        idx.extend(range(t0, t0 + block))
    n = len(idx)
    init = np.concatenate([w.init_positions() for w in ws])
    orc = oracle_py.OracleBank(0, n, ws[0].anchors, init_pos=init, n_threads=4)
    err = np.concatenate([w.err_est() for w in ws])
    out = np.zeros((steps, 3, n))
    for s in range(steps):
        orc.step_toa(np.concatenate([w.ranges_mm(s) for w in ws]), err, ws[0].dt_of(s))
        out[s] = orc.get_state()[0][:, :3].T
    return np.array(idx), out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_config4_sharded_equals_one_bank_and_oracle(world, tmp_path):
    if not has_gpu():
        pytest.skip("no GPU")
    got = _spawn(world, TOTAL, S, str(tmp_path))
    single = _unsharded(TOTAL, S)                                    # [S][3][TOTAL]
    for s in range(S):
        assert np.array_equal(got[f"epoch_{s}"], single[s]), f"per-epoch gather, epoch {s}"
        assert np.array_equal(got[f"trajectory_{s}"], single[s]), f"trajectory gather, epoch {s}"
    launch_keys = sorted(int(k.split("_")[1]) for k in got.files if k.startswith("launch_"))
    assert launch_keys == [4, 9]                                    # the last epoch of each 5-epoch launch
    for s in launch_keys:
        assert np.array_equal(got[f"launch_{s}"], single[s]), f"per-launch gather, epoch {s}"
    idx, orc = _oracle_sample(TOTAL, S)
    d = np.stack([got[f"epoch_{s}"][:, idx] for s in range(S)]) - orc
    rms = float(np.sqrt((d ** 2).sum(1).mean()))
    assert rms <= 1e-6, rms                                          # BASELINE bar; observed ~1e-15


@pytest.mark.gpu
def test_unequal_shards_on_gpu(tmp_path):
    """shard_range hands out shards that differ by one tag: the padded gather must still reassemble the batch."""
    if not has_gpu():
        pytest.skip("no GPU")
    total = 3 * 20000 + 2                                            # sizes 20001, 20001, 20000: one-tag-per-lane kernels
    got = _spawn(3, total, 6, str(tmp_path))
    single = _unsharded(total, 6)
    for s in range(6):
        assert got[f"epoch_{s}"].shape == (3, total)
        assert np.array_equal(got[f"epoch_{s}"], single[s])
        assert np.array_equal(got[f"trajectory_{s}"], single[s])
