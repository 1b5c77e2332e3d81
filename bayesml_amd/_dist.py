"""Row sharding of the sample matrix over one process per GPU.

The reference has no parallelism of any kind.  The data pass is linear in
the rows, so rank g owns a contiguous block of rows, every rank holds the K-sized state, and the
only exchange is ONE all-reduce(sum, f64) of the statistics block
``[ns | h | a | B]`` (K(2 + D + D^2) doubles) per VB iteration — RCCL over xGMI when the process
group's backend is "nccl", gloo in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class RowShard:
    """Describes which rows of the global sample matrix this process holds."""

    def __init__(self, group=None, native: bool = False, always: bool = False):
        """``always=True``: issue the collective even in a group of one rank (bench.py --force-dist exercises the RCCL
        path that way).
        ``native=True``: the per-iteration all-reduce of the statistics block goes through the library's own RCCL
        communicator (C ABI ``gmmvb_allreduce_stats``, bootstrapped over the process group) instead of
        ``torch.distributed.all_reduce``; everything else (row counts, sub-sample moments) stays on the group."""
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("torch.distributed is not initialised; call init_process_group first")
        self.group = group
        self.native = native
        self.always = always
        self._rccl = None
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.row_offset = 0
        self.global_rows = 0
        self.local_rows = 0

    def bind_rows(self, local_rows: int, device) -> "RowShard":
        """Exchange the local row counts once per update_posterior call (an all-gather of one int64)."""
        mine = torch.tensor([int(local_rows)], dtype=torch.int64, device=device)
        counts = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(counts, mine, group=self.group)
        counts = [int(c.item()) for c in counts]
        self.local_rows = int(local_rows)
        self.row_offset = sum(counts[:self.rank])
        self.global_rows = sum(counts)
        return self

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over ranks of a contiguous f64 tensor (the per-iteration collective)."""
        if not (self.world > 1 or self.always):
            return t
        if self.native and t.is_cuda:
            if self._rccl is None:
                from ._engine import RcclComm
                self._rccl = RcclComm(self.rank, self.world, t.device, group=self.group)
            return self._rccl.all_reduce_(t)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def close(self):
        """Release the library's RCCL communicator (``native=True``); the process group is the caller's."""
        if self._rccl is not None:
            self._rccl.close()
            self._rccl = None

    def local_indices(self, global_idx: torch.Tensor) -> torch.Tensor:
        """Rows of a global index list that this rank owns, as local row numbers."""
        sel = (global_idx >= self.row_offset) & (global_idx < self.row_offset + self.local_rows)
        return global_idx[sel] - self.row_offset


class RestartShard:
    """Restart-level parallelism for sample matrices small enough to be replicated: every
    process holds ALL rows; restart i of ``update_posterior`` runs on rank i mod world (each rank still draws every
    restart's random numbers, so the streams - and the result - are those of a single process), the winner is chosen
    with the reference's rule over the gathered lower bounds and its posterior is broadcast from its owner.
    No collective inside a VB iteration.

    One deviation from a single process: the reference keeps ``s_mats[k]`` of a component with ``ns[k] == 0`` from the
    previous pass - across restarts too (``_gaussianmixture.py:729``, "stale" S).  Here every rank only sees the chain of
    its own restarts, so for an EXACTLY empty component ``s_mats[k]`` (an attribute, not part of the posterior) can differ
    from the single-process value and between ranks."""

    restart_parallel = True
    row_offset = 0

    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("torch.distributed is not initialised; call init_process_group first")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.local_rows = self.global_rows = 0

    def bind_rows(self, local_rows: int, device) -> "RestartShard":
        self.local_rows = self.global_rows = int(local_rows)
        return self

    def all_reduce_(self, t):
        return t                      # rows are not sharded

    def local_indices(self, global_idx):
        return global_idx

    def owner(self, restart: int) -> int:
        return restart % self.world

    def gather(self, obj) -> list:
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def broadcast_(self, tensors, src: int):
        """``src`` is a rank of ``self.group``; torch.distributed wants the global rank."""
        gsrc = dist.get_global_rank(self.group, src) if self.group is not None else src
        for t in tensors:
            dist.broadcast(t, src=gsrc, group=self.group)


class SingleProcess:
    """The world_size == 1 stand-in with the same surface (no torch.distributed needed)."""

    rank, world, row_offset = 0, 1, 0

    def bind_rows(self, local_rows: int, device) -> "SingleProcess":
        self.local_rows = self.global_rows = int(local_rows)
        return self

    def all_reduce_(self, t):
        return t

    def local_indices(self, global_idx):
        return global_idx
