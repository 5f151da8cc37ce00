"""Multi-GPU fan-out of independent trajectories / chains: one process per GPU.

The hot path shards with no data-path collective: parameter samples (cfg 2/3/5) and MCMC chains
(cfg 4) are independent, sharing only read-only inputs (contact matrix, save grid, observations).
Each rank integrates a contiguous block of the batch on its own GPU; results stay sharded in HBM.
Collectives (RCCL over xGMI via ``torch.distributed``, backend "nccl"; "gloo" in CPU tests) are
used only AFTER the solve: a gather of small per-trajectory outputs (status, posterior draws) or
an all-reduce of ensemble summaries.  The reference has no counterpart (no
vmap/pmap/sharding code); chain fan-out there is numpyro's ``chain_method``.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import torch
import torch.distributed as dist


def world() -> tuple:
    """(rank, world_size) of the default process group, (0, 1) when not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(total: int, rank: int, world_size: int) -> tuple:
    """Contiguous block [lo, hi) of `total` items for `rank`; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass
class ShardedResult:
    """This rank's block of a sharded batched solve."""

    lo: int
    hi: int
    total: int
    local: object  # engine.BatchResult of trajectories [lo, hi)


def solve_sharded(model, y0, params, contact, t1, save_ts, *, solver: Optional[Callable] = None, **kw) -> ShardedResult:
    """Integrate this rank's block of the batch.  No communication happens here.

    ``params`` ([B, P]) and a batched ``y0`` ([B, D]) are the FULL arrays (host memory or any
    device); only rows [lo, hi) are moved to this rank's GPU.  ``solver`` defaults to
    :func:`dynode_amd.engine.solve_batch`.
    """
    if solver is None:
        from .engine import solve_batch as solver
    rank, size = world()
    B = params.shape[0]
    lo, hi = shard_bounds(B, rank, size)
    y0_local = y0[lo:hi] if getattr(y0, "ndim", 1) == 2 else y0
    local = solver(model, y0_local, params[lo:hi], contact, t1, save_ts, **kw)
    return ShardedResult(lo, hi, B, local)


def gather_rows(local: torch.Tensor, total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather per-trajectory rows from every rank onto ``dst`` in batch order (None elsewhere).

    Meant for small outputs -- status words, step counts, posterior draws (cfg 4: 1024 x 1000 x 2
    floats = 8 MB) -- not for full ensembles, which should stay sharded.
    Ragged shards are padded to the largest block for the collective and trimmed afterwards.
    """
    rank, size = world()
    if size == 1:
        return local
    sizes = [shard_bounds(total, r, size) for r in range(size)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(size)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def allreduce_ensemble_moments(ys_local: torch.Tensor) -> tuple:
    """Ensemble mean and variance over ALL ranks' trajectories, per (save time, state element).

    Sums are accumulated in float64 and all-reduced: 2 x n_save x D doubles per rank (cfg 5:
    800 KB) instead of gathering the 13 GB ensemble.
    """
    # float64 accumulation without a float64 (or squared) copy of the multi-GB ensemble: the
    # squares are formed in blocks of 1024 trajectories (a ~200 MB temporary for cfg 3)
    s1 = ys_local.sum(dim=0, dtype=torch.float64)
    s2 = torch.zeros_like(s1)
    for lo in range(0, ys_local.shape[0], 1024):
        s2 += ys_local[lo:lo + 1024].square().sum(dim=0, dtype=torch.float64)
    n = torch.tensor([float(ys_local.shape[0])], dtype=torch.float64, device=ys_local.device)
    rank, size = world()
    if size > 1:
        for t in (s1, s2, n):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    mean = s1 / n
    var = (s2 / n - mean * mean).clamp_min(0.0)
    return mean, var, int(n.item())
