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


class PoseGather:
    """All-gather of per-rank pose blocks [rows][t_local] -> [world][rows][t_pad], double-buffered.

    rows = 3 for one epoch's poses, K*3 for the poses of a K-epoch launch. `sizes` are the shard sizes of all
    ranks (shard_sizes()); they may differ: buffers are padded to t_pad = max(sizes), assemble() returns the
    [rows][sum(sizes)] array in global tag order. Works with any torch.distributed backend: nccl (= RCCL) on
    GPUs, gloo on CPU tensors (tests), gloo with GPU tensors staged through the host (rehearsals of the N > 1
    path on one card).
    """

    def __init__(self, t_local: int, device, dtype=None, overlap: bool = True, rows: int = 3, sizes=None):
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
        dtype = dtype or torch.float64
        self.local = [torch.zeros(rows, self.t_pad, dtype=dtype, device=device) for _ in range(2)]
        # concatenation form [world*rows][t_pad] (what every backend accepts); handed out as [world][rows][t_pad]
        self.flat = [torch.zeros(self.world * rows, self.t_pad, dtype=dtype, device=device) for _ in range(2)]
        self.full = [f.view(self.world, rows, self.t_pad) for f in self.flat]
        self.cuda = torch.device(device).type == "cuda"
        # gloo has no device collectives: stage through the host (rehearsals of the N>1 path on one card)
        self.host_staged = self.cuda and dist.is_initialized() and dist.get_backend() == "gloo"
        self.overlap = overlap and self.cuda and not self.host_staged
        if self.cuda:
            self.comm = torch.cuda.Stream(device=device) if self.overlap else None
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.done = [torch.cuda.Event() for _ in range(2)]
            self.used = [False, False]
        self.k = 0
        self.count = 0  # collectives issued

    @property
    def uniform(self) -> bool:
        """all shards the same size: nothing is padded anywhere"""
        return min(self.sizes) == self.t_pad

    def buffer(self):
        """Local block to fill for the next gather, [rows][t_pad] (columns >= t_local are padding); waits until
        the previous gather out of the same buffer has drained."""
        b = self.k & 1
        if self.overlap and self.used[b]:
            self.torch.cuda.current_stream().wait_event(self.done[b])
        return self.local[b]

    def gather(self):
        """Exchange the buffer handed out by buffer(); returns the [world][rows][t_pad] result tensor
        (valid after wait())."""
        b = self.k & 1
        self.k += 1
        self.count += 1
        if self.world == 1:
            self.full[b][0].copy_(self.local[b])
            return self.full[b]
        if self.overlap:
            self.ready[b].record(self.torch.cuda.current_stream())
            with self.torch.cuda.stream(self.comm):
                self.comm.wait_event(self.ready[b])
                self.dist.all_gather_into_tensor(self.flat[b], self.local[b])
                self.done[b].record(self.comm)
            self.used[b] = True
        elif self.host_staged:
            src = self.local[b].cpu()
            dst = self.torch.zeros(self.flat[b].shape, dtype=src.dtype)
            self.dist.all_gather_into_tensor(dst, src)
            self.flat[b].copy_(dst)
        else:
            self.dist.all_gather_into_tensor(self.flat[b], self.local[b])
        return self.full[b]

    def wait(self):
        if self.overlap:
            self.torch.cuda.current_stream().wait_stream(self.comm)

    def assemble(self, full):
        """[world][rows][t_pad] -> [rows][T_total] in global tag order (padding columns dropped)."""
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
                 stream=None):
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
            self.gather = PoseGather(self.T, device, rows=rows, sizes=self.sizes)

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
