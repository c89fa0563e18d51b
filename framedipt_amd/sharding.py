"""Sample sharding over the GPUs of one node.

Each (structure, sample index) item of a sampler dataset is an independent trajectory, so ranks take disjoint item
indices and never exchange data on the hot path; results are gathered on the host at the end.  One process per GPU
(``torch.distributed`` env contract); the only collective is the end-of-run gather / timing reduction.
"""
from __future__ import annotations

from typing import List


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin partition of ``range(n_items)``; ranks differ by at most one item."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    return list(range(rank, n_items, world))


def gather_results(local: dict, n_items: int, rank: int, world: int):
    """Host-side gather of ``{item_index: ndarray}`` to rank 0 (any backend; object gather)."""
    import torch.distributed as dist
    if world == 1:
        return dict(local)
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(local, bucket, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    if sorted(merged) != list(range(n_items)):
        raise RuntimeError("sample gather is incomplete")
    return merged


def max_over_ranks(value: float, device=None) -> float:
    """Timing reduction used by bench.py (MAX over ranks)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
