"""CPU tier: the N>1 path of bench.py -- row sharding, barrier and max-over-ranks timing --
exercised with two gloo ranks.  The compute leg is stubbed by the oracle (the checker), because
this tier has no GPU; what is under test is the partition/aggregation logic, which has no
data-path collective."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)
from bench import shard_rows  # noqa: E402


@pytest.mark.parametrize("total,world", [(10, 1), (10, 2), (11, 2), (1 << 20, 8), (7, 8), (0, 2)])
def test_shard_rows_partitions_exactly(total, world):
    spans = [shard_rows(total, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle
    orc = Oracle()
    total = 37
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (total, 960)).astype(np.float32)      # same on every rank
    lo, hi = shard_rows(total, world, rank)
    fin, tail = orc.imdct_batch(0, x[lo:hi], None)
    # bench.py's timing reduction: barrier, then MAX over ranks
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # gather only to let rank 0 verify coverage (NOT on the product's data path)
    parts = [None] * world
    dist.all_gather_object(parts, (lo, hi, fin, tail))
    if rank == 0:
        full_fin = np.concatenate([p[2] for p in parts])
        full_tail = np.concatenate([p[3] for p in parts])
        wf, wt = orc.imdct_batch(0, x, None)
        q.put((float(t.item()), bool(np.array_equal(full_fin, wf) and np.array_equal(full_tail, wt)),
               [(p[0], p[1]) for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    tmax, same, spans = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert tmax == 1.5                      # max over ranks, not rank 0's own time
    assert same                             # shards tile the batch exactly
    assert spans == [(0, 19), (19, 37)]
