"""Multi-GPU sharding of the tag batch.

Tags share nothing but the read-only anchor table, so the batch shards embarrassingly: rank g owns the
contiguous global tag range shard_range(T, G, g), keeps that state on its own GPU for good, and the only
exchange is ONE all-gather of poses per epoch (RCCL over xGMI; backend "nccl" is RCCL on ROCm). The
reference has no counterpart -- it runs one filter in one process (node_pos.cpp:176-181).

The gather is double-buffered on a side stream so that the exchange of epoch s overlaps the compute of
epoch s+1: xGMI is point-to-point (7 links per GPU), and a 65 536-tag f64 pose shard is only 1.5 MB, so
the collective is latency-, not bandwidth-bound; hiding it is what protects weak scaling.
"""
from __future__ import annotations

import os


def shard_range(n_tags_total: int, world: int, rank: int):
    """Contiguous [lo, hi) of global tag indices owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_tags_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


class PoseGather:
    """All-gather of per-rank pose shards [3][T_local] -> [world][3][T_local], double-buffered.

    Works with any torch.distributed backend (nccl/RCCL on GPUs, gloo on CPU for the tests). Equal
    shard sizes are required (weak scaling: every rank owns the same number of tags).
    """

    def __init__(self, t_local: int, device, dtype=None, overlap: bool = True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        dtype = dtype or torch.float64
        self.local = [torch.zeros(3, t_local, dtype=dtype, device=device) for _ in range(2)]
        # concatenation form [world*3][T_local] (what every backend accepts); handed out as [world][3][T_local]
        self.flat = [torch.zeros(self.world * 3, t_local, dtype=dtype, device=device) for _ in range(2)]
        self.full = [f.view(self.world, 3, t_local) for f in self.flat]
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

    def buffer(self):
        """Local pose buffer to fill for this epoch (waits until its previous gather has drained)."""
        b = self.k & 1
        if self.overlap and self.used[b]:
            self.torch.cuda.current_stream().wait_event(self.done[b])
        return self.local[b]

    def gather(self):
        """Exchange the buffer handed out by buffer(); returns the [world][3][T_local] result tensor
        (valid after wait())."""
        b = self.k & 1
        self.k += 1
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
