"""Multi-GPU sharding of the commitment path (one process per GPU, torch.distributed; the "nccl"
backend is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path shards along two axes; the prover's own sharding lives in the library
(`ProvingKey.set_sharding`: cq_pk_set_sharding / cq_pk_set_column_sharding, csrc/comm.hip), this module holds the same
rules for host code and tests:
  * the (scalar, base) index range of ONE multiexp -> `shard_range` + `sharded_multiexp`, whose only
    exchange is an all-gather of one 96-byte Jacobian partial per rank followed by a local EC sum
    (EC addition is not a reduction op RCCL offers, so "all-reduce" = all-gather + local sum);
  * a batch of independent column transforms -> `column_owner`: contiguous ranges of columns per rank (the same
    `shard_range` over the batch), outputs broadcast from their owner.
`backend` is any object with `best_multiexp(coeffs, bases) -> uint64[12]` and
`g1_sum(points uint64[m,12]) -> uint64[12]`; in production that is `GpuBackend(Context)`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of an n-term multiexp owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def column_owner(column: int, batch: int, world: int) -> int:
    """Owner rank of column `column` of a batch of `batch` independent column transforms: the rank whose
    `shard_range(batch, rank, world)` holds it (what `sharded_transform` in csrc/prover.hip uses)."""
    for r in range(world):
        lo, hi = shard_range(batch, r, world)
        if lo <= column < hi:
            return r
    raise IndexError(column)


class GpuBackend:
    def __init__(self, ctx):
        self.ctx = ctx

    def best_multiexp(self, coeffs: np.ndarray, bases: np.ndarray) -> np.ndarray:
        return self.ctx.best_multiexp(coeffs, bases)

    def g1_sum(self, points: np.ndarray) -> np.ndarray:
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
        out = np.zeros(12, dtype=np.uint64)
        self.ctx._chk(self.ctx.lib.cq_g1_sum(pts.ctypes.data, pts.shape[0], out.ctypes.data))
        return out


def sharded_multiexp(backend, coeffs_local: np.ndarray, bases_local: np.ndarray, group=None, device=None) -> np.ndarray:
    """Every rank passes ITS slice (see shard_range); every rank returns the full sum (Jacobian)."""
    import torch
    import torch.distributed as dist

    partial = backend.best_multiexp(coeffs_local, bases_local)
    world = dist.get_world_size(group)
    t = torch.from_numpy(partial.astype(np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)
    pts = np.stack([g.cpu().numpy().view(np.uint64) for g in gathered])
    return backend.g1_sum(pts)
