"""world_size-2 test of the multi-GPU plumbing on CPU (gloo): shards are disjoint and cover the batch, seeds are
global-id based, and the timing reduction is a MAX over ranks -- the same code path bench.py takes under torchrun."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from social_navigation_pyenvs_amd import scenarios as sc
from social_navigation_pyenvs_amd.sharding import max_over_ranks, shard_seed, sum_over_ranks, world_shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    first, count = world_shard(rank, world_size, worlds_per_gpu=8)
    S, goals, P, bounds = sc.hybrid_worlds(count, 5, "hsfm_farina", seed0=shard_seed(first), first_world=first)
    dist.barrier()
    slowest = max_over_ranks(0.010 * (rank + 1), dist)          # rank 1 is "slower"
    total = sum_over_ranks(float(count), dist)
    # the shards must differ (different seeds) and be reproducible from the global world id
    digest = float(np.round(S[:, :, 0:2].sum(), 6))
    gathered = [None] * world_size
    dist.all_gather_object(gathered, (first, count, digest))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, slowest, total, gathered))


def test_two_rank_sharding_and_timing_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, slowest, total, gathered in res:
        assert abs(slowest - 0.020) < 1e-12          # MAX over ranks
        assert total == 16.0                         # whole-job world count
        (f0, c0, d0), (f1, c1, d1) = gathered
        assert (f0, c0) == (0, 8) and (f1, c1) == (8, 8)   # contiguous, disjoint, covering
        assert d0 != d1
    # a world is a function of (seed, global id): the two shards together are the single-process batch, world for world
    S_all, _, _, _ = sc.hybrid_worlds(16, 5, "hsfm_farina", seed0=shard_seed(0))
    (f0, c0, d0), (f1, c1, d1) = res[0][3]
    assert d0 == float(np.round(S_all[:8, :, 0:2].sum(), 6)) and d1 == float(np.round(S_all[8:, :, 0:2].sum(), 6))


def test_strong_scaling_split_is_even_and_covering():
    for total, R in ((65536, 8), (10, 4), (7, 8)):
        spans = [world_shard(r, R, 0, total) for r in range(R)]
        assert sum(c for _, c in spans) == total
        pos = 0
        for f, c in spans:
            assert f == pos
            pos += c
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_single_process_is_identity():
    assert max_over_ranks(1.5) == 1.5 and sum_over_ranks(2.0) == 2.0
    assert torch.distributed.is_available()
