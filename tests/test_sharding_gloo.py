"""world_size-2 gloo test (CPU): disjoint sample shards, host gather, MAX timing reduction."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from framedipt_amd import sharding
    mine = sharding.shard_indices(n_items, rank, world)
    local = {i: np.full((3,), float(i)) for i in mine}
    merged = sharding.gather_results(local, n_items, rank, world)
    tmax = sharding.max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((sorted(merged), float(sum(v.sum() for v in merged.values())), tmax, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    from framedipt_amd import sharding
    assert sharding.shard_indices(5, 0, 2) == [0, 2, 4] and sharding.shard_indices(5, 1, 2) == [1, 3]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    keys, total, tmax, mine = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert keys == list(range(7)) and total == 3 * sum(range(7)) and tmax == 2.0 and mine == [0, 2, 4, 6]
