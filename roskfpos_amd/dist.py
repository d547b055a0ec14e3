"""Multi-GPU sharding of the tag batch.

Tags share nothing but the read-only anchor table, so the batch shards embarrassingly: rank g owns the
contiguous global tag range shard_range(T, G, g), keeps that state on its own GPU for good, and the only
exchange is the all-gather of poses (RCCL over xGMI; backend "nccl" is RCCL on ROCm). The reference has no
counterpart -- it runs one filter in one process (node_pos.cpp:176-181).

How often poses are exchanged is the caller's choice (ShardedReplay.gather_mode; SURVEY 8e: "once per step or
once per K steps"):
  "epoch"       one all-gather per ranging epoch -- what a live multi-tag node publishes at its 20 Hz tick
  "launch"      one all-gather per kernel launch, carrying the poses after the launch's LAST epoch (K = epochs
                per launch)
  "trajectory"  one all-gather per kernel launch carrying the poses after EVERY epoch of the launch
                ([K*3][T_local] per rank: fewer, larger collectives)
  "none"        no exchange
Every gather is double-buffered on a side stream so that the exchange of one epoch / launch overlaps the compute
of the next: xGMI is point-to-point (7 links per GPU), and a 131 072-tag f64 pose shard is only 3 MB, so a
collective is latency-, not bandwidth-bound; hiding it is what protects scaling.

Shards produced by shard_range differ by at most one tag; all_gather_into_tensor wants equal contributions, so
every rank's buffer is padded to the largest shard (at most one unused column per row) and assemble() drops the
padding again.
"""
from __future__ import annotations

import os

GATHER_MODES = ("epoch", "launch", "trajectory", "none")


def shard_range(n_tags_total: int, world: int, rank: int):
    """Contiguous [lo, hi) of global tag indices owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_tags_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_tags_total: int, world: int):
    return [hi - lo for lo, hi in (shard_range(n_tags_total, world, r) for r in range(world))]


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def default_engine(device) -> str:
    """Which implementation exchanges the poses: "cabi" = the library's own RCCL communicator behind the C ABI
    (kfpos_allgather_poses: what a C++ node uses too), "torch" = torch.distributed's all_gather_into_tensor. The C ABI
    needs one GPU per rank and RCCL, so gloo runs (CPU tests, rehearsals of several ranks on one card) use torch.
    KFPOS_GATHER_ENGINE=torch|cabi overrides."""
    import torch
    import torch.distributed as dist
    forced = os.environ.get("KFPOS_GATHER_ENGINE")
    if forced in ("torch", "cabi"):
        return forced
    if torch.device(device).type != "cuda":
        return "torch"
    if dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() != "nccl":
        return "torch"
    return "cabi" if (dist.is_initialized() and dist.get_world_size() > 1) else "torch"


def _consensus_device(device):
    """where the small tensors live on which the ranks agree (all-reduce): the device for RCCL, the host for gloo"""
    import torch.distributed as dist
    return "cpu" if dist.get_backend() == "gloo" else device


def make_pose_gather(t_local: int, device, rows: int = 3, sizes=None, engine: str = None):
    """PoseGather with the engine default_engine() picks -- and, when that is the C ABI, a safety net for its first
    contact with a machine: the communicator is created and checked with one predictable gather; if ANY rank fails
    either step, every rank falls back to torch.distributed's collective (agreed through an all-reduce), and the
    returned object says which engine it ended up with (.engine, .fallback_reason)."""
    import torch
    import torch.distributed as dist
    want = engine or default_engine(device)
    if want != "cabi" or not dist.is_initialized() or dist.get_world_size() == 1:
        g = PoseGather(t_local, device, rows=rows, sizes=sizes, engine=want)
        g.fallback_reason = None
        g.calibration = None
        return g
    g, why = None, None
    try:
        g = PoseGather(t_local, device, rows=rows, sizes=sizes, engine="cabi")
        if not g.self_check():
            why = "the check gather returned something else than the pattern sent"
    except Exception as e:  # noqa: BLE001 -- whatever went wrong, the other ranks have to hear about it
        why = f"{type(e).__name__}: {e}"
    flag = torch.tensor([0 if why else 1], dtype=torch.int32, device=_consensus_device(device))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag[0]) == 1:
        g.fallback_reason = None
        g.calibration = _pick_algorithm(g, device)
        return g
    if g is not None:
        g.close()
    g = PoseGather(t_local, device, rows=rows, sizes=sizes, engine="torch")
    g.fallback_reason = why or "another rank could not use the C-ABI gather"
    g.calibration = None
    return g


def _pick_algorithm(g, device, reps: int = 6):
    """Which way the blocks should travel on THIS machine at THIS block size: RCCL's all-gather, or every rank sending
    its block straight to every other one (include/kfpos.h: KFPOS_GATHER_DIRECT; on MI355X each pair of GPUs has an xGMI
    link of its own). Both are checked with the predictable pattern and timed -- `reps` back-to-back gathers, the
    slowest rank counts -- and the faster one that passed stays selected. Every rank reaches the same verdict (the
    numbers are all-reduced). KFPOS_GATHER_ALGO=direct|collective pins the choice instead."""
    import time
    import torch
    import torch.distributed as dist
    from . import capi
    forced = os.environ.get("KFPOS_GATHER_ALGO")
    if forced in ("direct", "collective"):
        g.comm.set_algorithm(capi.GATHER_DIRECT if forced == "direct" else capi.GATHER_COLLECTIVE)
        return {"picked": forced, "why": "KFPOS_GATHER_ALGO"}
    out = {}
    for name, algo in (("collective", capi.GATHER_COLLECTIVE), ("direct", capi.GATHER_DIRECT)):
        ok, us = 1, float("inf")
        try:
            g.comm.set_algorithm(algo)
            ok = 1 if g.self_check() else 0
            if ok:
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(reps):
                    g.gather()
                g.wait()
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) * 1e6 / reps
        except Exception:  # noqa: BLE001
            ok = 0
        v = torch.tensor([float(ok), -us if ok else float("-inf")], dtype=torch.float64, device=_consensus_device(device))
        dist.all_reduce(v, op=dist.ReduceOp.MIN)   # ok on every rank; the slowest rank's time
        out[name] = {"ok": bool(v[0] == 1.0), "us_per_gather": (-float(v[1])) if v[0] == 1.0 else None}
    good = [n for n in out if out[n]["ok"]]
    pick = min(good, key=lambda n: out[n]["us_per_gather"]) if good else "collective"
    g.comm.set_algorithm(capi.GATHER_DIRECT if pick == "direct" else capi.GATHER_COLLECTIVE)
    out["picked"] = pick
    out["block_bytes_per_rank"] = g.rows * g.t_pad * 8
    return out


class PoseGather:
    """All-gather of per-rank pose blocks [rows][t_local] -> every rank holds [rows][T_total], double-buffered.

    rows = 3 for one epoch's poses, K*3 for the poses of a K-epoch launch. `sizes` are the shard sizes of all
    ranks (shard_sizes()); they may differ by one tag.

    engine "cabi": kfpos_allgather_poses of the library (RCCL communicator, side stream, padding and assembly inside
    the call; include/kfpos.h) -- the path a C++ node takes. gather() returns the assembled [rows][T_total] tensor.
    engine "torch": torch.distributed with any backend -- nccl (= RCCL) on GPUs, gloo on CPU tensors (tests), gloo
    with GPU tensors staged through the host (rehearsals of the N > 1 path on one card). Buffers are padded to
    t_pad = max(sizes); gather() returns [world][rows][t_pad] and assemble() cuts the padding away.
    Either way: buffer() -> fill -> gather() -> wait() -> assemble(result) is [rows][T_total] in global tag order.
    """

    def __init__(self, t_local: int, device, dtype=None, overlap: bool = True, rows: int = 3, sizes=None,
                 engine: str = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.sizes = list(sizes) if sizes is not None else [t_local] * self.world
        if len(self.sizes) != self.world or self.sizes[self.rank] != t_local:
            raise ValueError(f"sizes {self.sizes} do not describe this rank's shard of {t_local} tags")
        self.t_local, self.rows = t_local, rows
        self.t_pad = max(self.sizes)
        self.total = sum(self.sizes)
        dtype = dtype or torch.float64
        self.cuda = torch.device(device).type == "cuda"
        self.engine = engine or default_engine(device)
        self.k = 0
        self.count = 0  # collectives issued
        self.comm = None
        if self.engine == "cabi":
            if not self.cuda or dtype != torch.float64:
                raise ValueError("the C-ABI gather exchanges f64 poses resident in HBM")
            self._init_cabi(device)
            return
        self.local = [torch.zeros(rows, self.t_pad, dtype=dtype, device=device) for _ in range(2)]
        # concatenation form [world*rows][t_pad] (what every backend accepts); handed out as [world][rows][t_pad]
        self.flat = [torch.zeros(self.world * rows, self.t_pad, dtype=dtype, device=device) for _ in range(2)]
        self.full = [f.view(self.world, rows, self.t_pad) for f in self.flat]
        # gloo has no device collectives: stage through the host (rehearsals of the N>1 path on one card)
        self.host_staged = self.cuda and dist.is_initialized() and dist.get_backend() == "gloo"
        self.overlap = overlap and self.cuda and not self.host_staged
        if self.cuda:
            self.side = torch.cuda.Stream(device=device) if self.overlap else None
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.done = [torch.cuda.Event() for _ in range(2)]
            self.used = [False, False]

    # ---- engine "cabi" ----
    def _init_cabi(self, device):
        from . import capi
        torch, dist = self.torch, self.dist
        dev = torch.device(device)
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        if self.sizes != shard_sizes(self.total, self.world):
            raise ValueError("the C-ABI gather shards by kfpos_shard_range; these sizes are something else")
        uid = [capi.comm_unique_id() if self.rank == 0 else None]
        if self.world > 1:  # ship rank 0's ncclUniqueId over the process group that is already up
            dist.broadcast_object_list(uid, src=0, device=_consensus_device(dev))
        self.comm = capi.KfposComm(self.world, self.rank, uid[0], device=index)
        lo, hi = self.comm.set_total(self.total)
        assert hi - lo == self.t_local and (lo, hi) == shard_range(self.total, self.world, self.rank)
        # tight [rows][t_local] blocks to fill when the caller has no contiguous source of its own
        self.local = [torch.zeros(self.rows, self.t_local, dtype=torch.float64, device=device) for _ in range(2)]
        self.out = [torch.zeros(self.rows, self.total, dtype=torch.float64, device=device) for _ in range(2)]

    def self_check(self) -> bool:
        """One gather of a pattern every rank can predict (global tag index + 1000 x row): True when this rank received
        exactly that. Collective; leaves the double-buffer parity as it found it (two gathers)."""
        torch = self.torch
        lo = sum(self.sizes[:self.rank])
        dev = self.local[0].device
        cols = torch.arange(lo, lo + self.t_local, dtype=torch.float64, device=dev)
        rows = torch.arange(self.rows, dtype=torch.float64, device=dev)[:, None] * 1000.0
        want = torch.arange(self.total, dtype=torch.float64, device=dev)[None, :] + rows
        ok = True
        for _ in range(2):
            buf = self.buffer()
            buf[:, :self.t_local].copy_(cols[None, :] + rows)
            got = self.gather()
            self.wait()
            if self.cuda:
                torch.cuda.current_stream().synchronize()
            ok = ok and bool(torch.equal(self.assemble(got), want))
        return ok

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    @property
    def uniform(self) -> bool:
        """all shards the same size: nothing is padded anywhere"""
        return min(self.sizes) == self.t_pad

    def buffer(self):
        """Local block to fill for the next gather, [rows][t_pad] (columns >= t_local are padding; engine "cabi":
        [rows][t_local]); waits until the previous gather out of the same buffer has drained."""
        b = self.k & 1
        if self.engine == "torch" and self.overlap and self.used[b]:
            self.torch.cuda.current_stream().wait_event(self.done[b])
        return self.local[b]

    def gather(self, src=None, rows=None):
        """Exchange the buffer handed out by buffer() -- or, engine "cabi", a contiguous [rows][t_local] device block
        of the caller's (no staging copy: the library packs it on the current stream before it returns). Returns the
        result tensor, valid after wait(): [world][rows][t_pad] (torch) or the assembled [rows][T_total] (cabi)."""
        b = self.k & 1
        self.k += 1
        self.count += 1
        if self.engine == "cabi":
            rows = self.rows if rows is None else rows
            if src is None:
                src = self.local[b]
            elif not (src.is_contiguous() and src.numel() == rows * self.t_local):
                raise ValueError("src must be a contiguous [rows][t_local] block")
            # the pack runs on torch's current stream, behind whatever produced src; the collective and the assembly on
            # the communicator's side stream; the third gather in flight waits for the first (inside the call)
            self.comm.allgather(self.out[b], pos_local=src, rows=rows,
                                stream=self.torch.cuda.current_stream().cuda_stream)
            return self.out[b][:rows]
        if src is not None:
            self.local[b][:src.shape[0], :self.t_local].copy_(src.reshape(-1, self.t_local), non_blocking=True)
        if self.world == 1:
            self.full[b][0].copy_(self.local[b])
            return self.full[b]
        if self.overlap:
            self.ready[b].record(self.torch.cuda.current_stream())
            with self.torch.cuda.stream(self.side):
                self.side.wait_event(self.ready[b])
                self.dist.all_gather_into_tensor(self.flat[b], self.local[b])
                self.done[b].record(self.side)
            self.used[b] = True
        elif self.host_staged:
            src_h = self.local[b].cpu()
            dst = self.torch.zeros(self.flat[b].shape, dtype=src_h.dtype)
            self.dist.all_gather_into_tensor(dst, src_h)
            self.flat[b].copy_(dst)
        else:
            self.dist.all_gather_into_tensor(self.flat[b], self.local[b])
        return self.full[b]

    def wait(self):
        """the current stream waits for the gathers issued so far"""
        if self.engine == "cabi":
            self.comm.wait(self.torch.cuda.current_stream().cuda_stream)
        elif self.overlap:
            self.torch.cuda.current_stream().wait_stream(self.side)

    def assemble(self, full):
        """gather()'s result -> [rows][T_total] in global tag order (padding columns dropped)."""
        if full.dim() == 2:  # engine "cabi": assembled inside the call
            return full
        if self.uniform:
            return full.permute(1, 0, 2).reshape(self.rows, self.world * self.t_pad)
        return self.torch.cat([full[r, :, :n] for r, n in enumerate(self.sizes)], dim=1)


def device_trace(torch, w, n_steps: int, device, with_imu: bool, real):
    """Device-resident trace of one shard (w: synth.Workload of the shard's tags): ranges [S][A][T] int32,
    accel [S][3][T] and cov [9][T] of element type `real` (9-state only), err [A][T] `real`, dts (S,) host array,
    traj [S][3][T] f64 (the pose after every epoch, written by the step kernels)."""
    import numpy as np
    T, A = w.n_tags, w.n_anchors
    tr = {"ranges": torch.empty((n_steps, A, T), dtype=torch.int32, device=device), "dts": np.zeros(n_steps)}
    if with_imu:
        tr["accel"] = torch.empty((n_steps, 3, T), dtype=torch.float32 if real == np.float32 else torch.float64,
                                  device=device)
        tr["cov"] = torch.from_numpy(np.ascontiguousarray(w.accel_cov(real).T)).to(device)
    for s in range(n_steps):
        tr["ranges"][s].copy_(torch.from_numpy(np.ascontiguousarray(w.ranges_mm(s).T)))
        if with_imu:
            tr["accel"][s].copy_(torch.from_numpy(np.ascontiguousarray(w.accel(s, real).T)))
        tr["dts"][s] = w.dt_of(s)
    tr["err"] = torch.from_numpy(np.ascontiguousarray(w.err_est(real).T)).to(device)
    tr["traj"] = torch.zeros((n_steps, 3, T), dtype=torch.float64, device=device)
    return tr


class ShardedReplay:
    """One rank's shard of a tag bank replaying a trace that is resident in HBM, plus the pose exchange.

    bank: capi.KfposBank of this rank's t_local tags. trace tensors (device): ranges [S][A][T] int32,
    err [A][T], optional accel [S][3][T] / cov [9][T] (9-state), traj [S][3][T] f64 (the pose after every
    epoch, written by the step kernels); dts: host array (S,). Used by bench.py and by the shard-equivalence
    tests, so both exercise the same code.
    """

    def __init__(self, bank, n_tags_total: int, device, gather_mode: str = "launch", epochs_per_launch: int = 25,
                 stream=None, engine: str = None, gather: "PoseGather" = None):
        import torch
        import torch.distributed as dist
        if gather_mode not in GATHER_MODES:
            raise ValueError(f"gather_mode must be one of {GATHER_MODES}")
        self.torch = torch
        self.bank, self.device, self.mode = bank, device, gather_mode
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.sizes = shard_sizes(n_tags_total, self.world)
        self.T = self.sizes[self.rank]
        if bank.T != self.T:
            raise ValueError(f"bank holds {bank.T} tags, this rank's shard is {self.T}")
        self.E = max(1, min(int(epochs_per_launch), 128))
        self.stream = stream
        self.launches = 0
        self.gather = None
        if gather_mode != "none":
            rows = 3 * self.E if gather_mode == "trajectory" else 3
            if gather is not None:  # a communicator is worth keeping: the caller shares one between replays
                if gather.rows != rows or gather.sizes != self.sizes:
                    raise ValueError("the PoseGather passed in was made for another block shape")
                self.gather = gather
            else:
                self.gather = make_pose_gather(self.T, device, rows=rows, sizes=self.sizes, engine=engine)

    def run(self, trace, s0: int, n: int, on_gathered=None):
        """Epochs [s0, s0 + n) of `trace` (dict: ranges, err, dts, traj, and accel / cov for the 9-state filter).
        on_gathered(first_epoch, n_epochs, full): called with every gathered result [world][rows][t_pad] after it
        has landed (tests; forces a wait per gather, so not for timing)."""
        ranges, err, dts, traj = trace["ranges"], trace["err"], trace["dts"], trace["traj"]
        accel, cov = trace.get("accel"), trace.get("cov")
        A, T, g = ranges.shape[1], self.T, self.gather
        s, end = s0, s0 + n
        while s < end:
            m = min(self.E, end - s)
            self.bank.run_trace_dev(m, ranges[s], A * T, err, 0, dts[s:s + m],
                                    accel=None if accel is None else accel[s], stride_accel=3 * T,
                                    cov=cov, stride_cov=0, trajectory=traj[s], stream=self.stream)
            self.launches += 1
            if g is not None:
                if self.mode == "epoch":
                    blocks = [(s + k, 1) for k in range(m)]
                elif self.mode == "launch":
                    blocks = [(s + m - 1, 1)]
                else:
                    blocks = [(s, m)]
                for first, cnt in blocks:
                    if g.engine == "cabi":  # the trajectory block itself is the contiguous [3 cnt][T] source
                        full = g.gather(src=traj[first:first + cnt], rows=3 * cnt)
                    else:
                        buf = g.buffer()
                        buf[:3 * cnt, :T].copy_(traj[first:first + cnt].reshape(3 * cnt, T), non_blocking=True)
                        full = g.gather()
                    if on_gathered is not None:
                        g.wait()
                        if g.cuda:
                            self.torch.cuda.current_stream().synchronize()
                        on_gathered(first, cnt, full)
            s += m
        if g is not None:
            g.wait()
