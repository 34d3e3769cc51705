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


def _bench_worker(rank, world_size, port, q):
    """bench.py's own argument / shard / seed plumbing under a 2-rank gloo group (everything of main() that needs no GPU)."""
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_RANK=str(rank))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    args = bench.parse(["--gpus", "2", "--steps", "5", "--warmup", "1", "--worlds", "12", "--agents", "5"])
    weak = bench.host_worlds(bench.workload_spec(args), rank, world_size)
    args5 = bench.parse(["--gpus", "2", "--total-worlds", "21", "--agents", "8", "--model", "hsfm_farina", "--scenario", "circle",
                         "--walls", "--static", "3"])
    strong = bench.host_worlds(bench.workload_spec(args5), rank, world_size)
    rec = dict(rank=rank, weak=(weak["first"], weak["W"], float(weak["S"][:, :, 0:2].sum()), weak["respawn_worlds"].tolist()),
               strong=(strong["first"], strong["W"], float(strong["S"][:, :, 0:2].sum())),
               scaling=("strong" if args5.total_worlds is not None else "weak", "strong" if args.total_worlds is not None else "weak"),
               keys=[bench.spec_key(o) for o in bench.other_config_specs(args)],
               cfg5_shard=bench.shard_of([o for o in bench.other_config_specs(args) if o["name"] == "cfg5"][0], rank, world_size))
    out = [None] * world_size
    dist.all_gather_object(out, rec)
    slow = max_over_ranks(0.5 + rank, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out, slow))


def test_bench_argument_and_seed_plumbing_two_ranks():
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, recs, slow = res[0]
    assert slow == 1.5
    r0, r1 = sorted(recs, key=lambda r: r["rank"])
    # weak scaling: 12 worlds per rank, contiguous global ids, hybrid parity by GLOBAL id
    assert r0["weak"][:2] == (0, 12) and r1["weak"][:2] == (12, 12)
    assert r0["weak"][3] == [i % 2 for i in range(12)] and r1["weak"][3] == [i % 2 for i in range(12, 24)]
    # strong scaling: 21 worlds split 11 + 10, and the union is the single-process batch world for world
    assert r0["strong"][:2] == (0, 11) and r1["strong"][:2] == (11, 10)
    assert r0["scaling"] == ("strong", "weak")
    one = bench.host_worlds(bench.workload_spec(bench.parse(["--total-worlds", "21", "--agents", "8", "--model", "hsfm_farina",
                                                             "--scenario", "circle", "--walls", "--static", "3"])), 0, 1)
    assert abs(float(one["S"][:11, :, 0:2].sum()) - r0["strong"][2]) < 1e-9 and abs(float(one["S"][11:, :, 0:2].sum()) - r1["strong"][2]) < 1e-9
    # BASELINE configs[4]: 65536 worlds over the ranks, 32768 each at two ranks
    assert r0["cfg5_shard"] == (0, 32768) and r1["cfg5_shard"] == (32768, 32768)
    # (cfg5 -- the 65536-world job, strong split -- and cfg5_shard -- 8192 worlds per GPU, weak -- read the same PMC entry: the same kernel build
    #  on the same kind of worlds)
    assert r0["keys"] == ["sfm_helbing_10_circle", "orca_25_circle_first20", "orca_25_circle_dense", "orca_25_circle_first20_fma", "orca_25_circle_dense_fma", "hsfm_farina_50_circle_walls_static",
                          "hsfm_farina_50_circle_walls_static", "hsfm_new_moussaid_25_hybrid", "hsfm_new_guo_25_hybrid", "hsfm_farina_25_hybrid_robot", "hsfm_farina_30_hybrid", "hsfm_farina_15_hybrid", "hsfm_farina_40_hybrid", "hsfm_farina_25_hybrid_peragent"]


class _HostOnlyEnv:
    """Stand-in for BatchedSocialNavGym on a box without a GPU: the worlds come from the package's host generators (functions of the
    test case = global world id), a step moves every human by the action of its world -- enough to check that the sharded wrapper
    hands every rank the right test cases, the right action rows and puts gathered shards back in world order."""

    def __init__(self, config, n_worlds, **kw):
        self.W, self.n = int(n_worlds), 5

    def reset(self, phase="test", first_case=0, **kw):
        pos, yaw, g = sc.circular_crossing(self.W, self.n, 7.0, 1000, first_world=first_case)
        self.S = sc.make_states(pos, yaw, g).astype(np.float32)
        return self.observe()

    def observe(self):
        return self.S[:, :, [0, 1, 3, 4, 8]].copy()

    def step(self, actions):
        self.S[:, :, 0:2] += np.asarray(actions, np.float32)[:, None, :] * 0.25
        return self.observe(), np.zeros(self.W, np.float32), np.zeros(self.W, bool), np.zeros(self.W, bool), np.zeros(self.W, np.int32)


def _sharded_worker(rank, world_size, port, q, total):
    from social_navigation_pyenvs_amd.social_gym.sharded_gym import ShardedBatchedSocialNavGym

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    env = ShardedBatchedSocialNavGym(None, total, dist=dist, env_factory=_HostOnlyEnv)
    obs = env.reset(phase="test", first_case=3)
    full0 = env.gather(obs)                                        # every rank gets the whole batch, in world order
    actions = np.stack([np.linspace(-1, 1, total), np.linspace(1, -1, total)], -1).astype(np.float32)   # one row per GLOBAL world
    obs1, *_ = env.step(actions)                                   # each rank takes its own rows: no communication
    on_learner = env.gather(torch.as_tensor(obs1), dst=0)          # tensors in -> tensors out, learner rank only
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, (env.first, env.W), full0, None if on_learner is None else on_learner.numpy()))


def test_sharded_gym_gathered_observations_equal_the_single_process_batch():
    """ShardedBatchedSocialNavGym under gloo with two ranks and a RAGGED split (21 worlds = 11 + 10): the gathered observations
    are the single-process batch world for world, before and after a step driven by one global action array; the step itself
    is collective-free (the only collective is gather's all_gather)."""
    total = 21
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, total)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = _HostOnlyEnv(None, total)
    ref0 = one.reset(first_case=3)
    actions = np.stack([np.linspace(-1, 1, total), np.linspace(1, -1, total)], -1).astype(np.float32)
    ref1 = one.step(actions)[0]
    assert res[0][1] == (0, 11) and res[1][1] == (11, 10)
    for rank, _, full0, learner in res:
        np.testing.assert_array_equal(full0, ref0)
        if rank == 0:
            np.testing.assert_array_equal(learner, ref1)
        else:
            assert learner is None
