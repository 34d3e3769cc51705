"""World sharding across the GPUs of one node.

Worlds are independent (no exchange step on the data path): rank r of R owns a contiguous
block of world ids and its own seed range; the only collectives are the timing barrier and a MAX over
ranks of the elapsed time (bench.py) -- RCCL on the GPU box (backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

import os


def env_rank():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def world_shard(rank: int, world_size: int, worlds_per_gpu: int, total_worlds: int | None = None):
    """Contiguous block of global world ids owned by `rank`.

    Weak scaling (total_worlds None): every rank owns `worlds_per_gpu` worlds.  Strong scaling: `total_worlds`
    are split as evenly as possible (the first `total % R` ranks get one more).  Returns (first_id, count)."""
    if total_worlds is None:
        return rank * worlds_per_gpu, worlds_per_gpu
    base, extra = divmod(total_worlds, world_size)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def shard_seed(first_world_id: int, base_seed: int = 1000) -> int:
    """Seed for a rank's worlds: the SAME base seed on every rank.  scenarios.* key every draw by (seed, global world
    id) -- pass `first_world=first_world_id` to them -- so a world's scenario does not depend on the shard boundaries or
    on how many GPUs the batch is spread over.  (`first_world_id` is accepted for symmetry and deliberately unused.)"""
    return base_seed


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """MAX all-reduce of a scalar (the timed region of the slowest rank); identity without a process group."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
