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


# ---------------------------------------------------------------------------------------------------------------------
# Per-sample random streams ("Caveat").  The reference draws x_T and every reverse step's noise from
# ONE global legacy np.random stream, sample after sample.  Sharded runs restart that stream per sample at `seed + item`,
# draw x_T (through the dataset item) and then the whole noise tape of the sample's trajectory in the reference's per-step
# order: a sample's inputs - and with batch-position-independent kernels its whole trajectory - do not depend on the world
# size, the rank that runs it or the batch it rides in.
def seeded_item(dataset, item: int, seed: int, diffuser, num_t: int, min_t: float):
    """``dataset[item]`` and its noise tape drawn from ``np.random.seed(seed + item)``.
    Returns (name_or_length, sample_i, feats [1,N,...], (z_rot, z_trans) [n_noisy,1,N,3] float64)."""
    import numpy as np

    from .inference import draw_noise_tape
    np.random.seed(seed + item)
    name, sample_i, feats = dataset[item]
    n = feats["rigids_t"].shape[1]
    n_noisy = int(np.sum(np.linspace(min_t, 1.0, num_t)[::-1] > min_t))
    return name, sample_i, feats, draw_noise_tape(diffuser, n_noisy, 1, n)


def stack_items(items):
    """Batch of equally sized samples from ``seeded_item`` tuples: (feats [B,N,...], (z_rot, z_trans) [n,B,N,3])."""
    import numpy as np
    import torch
    feats = {k: torch.cat([it[2][k] for it in items], dim=0) for k in items[0][2]}
    tape = tuple(np.concatenate([it[3][j] for it in items], axis=1) for j in range(2))
    return feats, tape


def batches_by_length(lengths, max_batch: int):
    """Group local item positions into batches of equal N (at most ``max_batch`` samples each), in item order."""
    out, cur = [], []
    for pos, n in enumerate(lengths):
        if cur and (lengths[cur[0]] != n or len(cur) == max_batch):
            out.append(cur)
            cur = []
        cur.append(pos)
    if cur:
        out.append(cur)
    return out
