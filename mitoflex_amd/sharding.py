"""Input sharding for the multi-GPU path.

Reads (pairs) are independent units: each rank / device takes one contiguous chunk of whole
pairs, the bait table is replicated, every rank writes its own range of the result, and the only
cross-rank step is a host-side sum of counts.  No RCCL collective is introduced; when ranks run
as separate processes (bench.py under torch.distributed.run) the sum goes through whatever
process group the caller has (gloo on CPU tensors is enough).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of units (pairs for PE, reads for SE) owned by `rank`; sizes differ by
    at most one; concatenating ranks 0..world-1 restores the input order."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    return n_units * rank // world, n_units * (rank + 1) // world


def reduce_counts(kept: int, total: int, dist=None) -> Tuple[int, int]:
    """Sum (kept, total) over ranks.  `dist` is torch.distributed (initialised) or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return kept, total
    import torch
    t = torch.tensor([kept, total], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t[0]), int(t[1])
