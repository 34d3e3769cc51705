"""``ShardedBatchedSocialNavGym``: a vectorised Gym of ``total_worlds`` worlds spread over the ranks of a ``torch.distributed``
job -- one process per GPU (``torchrun``), SURVEY.md §8(e).

Worlds are independent units: rank r of R owns the contiguous block of global world ids ``sharding.world_shard`` gives it and
steps them in its own ``BatchedSocialNavGym`` on its own GPU.  World w of the job is test case ``first_case + w`` of the phase
whatever R is -- and its k-th auto-reset draws seed ``s + w + k * total_worlds`` whatever R is (``seed_stride``) -- so the union of
the shards IS the single-process batch, world for world, episode after episode.  **The step has no collective**: reset,
step, step_device, imitation_learning_step and lookahead_device act on the local shard only.  The one optional exchange is
``gather`` -- an ``all_gather`` (RCCL on the GPU box: backend "nccl"; gloo in the CPU tests) of per-world tensors such as the
observations ``[W/R, N, 5]``, once per Gym step, for a learner that wants the whole batch on one rank; ``scatter_actions`` is
its inverse on the learner's side (every rank slices its own rows out of the full action array -- no communication).
"""
from __future__ import annotations

import numpy as np

from ..sharding import env_rank, world_shard


class ShardedBatchedSocialNavGym:
    def __init__(self, config, total_worlds: int, *, dist=None, rank=None, world_size=None, env_factory=None, **env_kw):
        """``dist``: an initialised ``torch.distributed`` module (None: single process).  ``rank`` / ``world_size`` default to the
        process group's (or the torchrun environment's).  ``env_factory(config, n_worlds, **env_kw)`` builds the local
        environment (default ``BatchedSocialNavGym``); the device is whatever the process selected (one rank per GPU:
        ``torch.cuda.set_device(LOCAL_RANK)`` / ``_lib.set_device`` before constructing)."""
        self.dist = dist if (dist is not None and dist.is_available() and dist.is_initialized()) else None
        if rank is None or world_size is None:
            if self.dist is not None:
                rank, world_size = self.dist.get_rank(), self.dist.get_world_size()
            else:
                r, _, ws = env_rank()
                rank, world_size = (r, ws) if rank is None and world_size is None else (rank or 0, world_size or 1)
        self.rank, self.world_size, self.total_worlds = int(rank), int(world_size), int(total_worlds)
        self.first, self.W = world_shard(self.rank, self.world_size, 0, self.total_worlds)
        # (first id, count) of every rank: what gather() needs to lay ragged shards out in world order
        self.shards = [world_shard(r, self.world_size, 0, self.total_worlds) for r in range(self.world_size)]
        if env_factory is None:
            from .social_nav_gym import BatchedSocialNavGym as env_factory
        self.env = env_factory(config, self.W, **env_kw)
        # a finished world's seed moves on by the worlds of the WHOLE job (cs_gym_book.seed_stride): global world w walks
        # s + w + k * total_worlds whatever the number of ranks, so the union of the shards stays the single-process batch across
        # auto-resets too (with the local W as the stride, rank 0's second episodes would be rank 1's first ones)
        self.env.seed_stride = self.total_worlds

    # ------------------------------------------------------------------ the local shard: no communication
    def reset(self, phase="test", first_case=0, **kw):
        """Worlds ``first_case + first .. first_case + first + W`` of ``phase``: global world w is the same test case on any rank count."""
        return self.env.reset(phase=phase, first_case=first_case + self.first, **kw)

    def step(self, actions):
        return self.env.step(self.local_rows(actions))

    def step_device(self, actions, **kw):
        return self.env.step_device(self.local_rows(actions), **kw)

    def __getattr__(self, name):   # everything else (observe, imitation_learning_step, lookahead_device, cw, n, ...) is the local env's
        return getattr(self.env, name)

    def local_rows(self, x):
        """This rank's rows of a per-world array given either for the shard ([W, ...]) or for the whole job ([total_worlds, ...])."""
        if hasattr(x, "shape") and len(x.shape) >= 1 and x.shape[0] == self.total_worlds and self.total_worlds != self.W:
            return x[self.first:self.first + self.W]
        return x

    scatter_actions = local_rows

    # ------------------------------------------------------------------ the one optional exchange
    def gather(self, x, dst=None):
        """``all_gather`` of a per-world array / tensor of the local shard ([W, ...]) into the whole job's ([total_worlds, ...]) in
        global world order.  Returns it on every rank (``dst=None``) or on rank ``dst`` only (None elsewhere).  numpy in ->
        numpy out (gloo / CPU tensors); CUDA tensor in -> CUDA tensor out (RCCL).  Ragged shards are padded to the largest one
        for the collective and trimmed afterwards."""
        import torch

        is_np = isinstance(x, np.ndarray)
        t = torch.as_tensor(x)
        if self.dist is None or self.world_size == 1:
            return x
        wmax = max(c for _, c in self.shards)
        if t.shape[0] < wmax:
            pad = torch.zeros((wmax - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], 0)
        parts = [torch.empty_like(t) for _ in range(self.world_size)]
        self.dist.all_gather(parts, t.contiguous())
        if dst is not None and self.rank != dst:
            return None
        full = torch.cat([p[:c] for p, (_, c) in zip(parts, self.shards)], 0)
        return full.cpu().numpy() if is_np else full
