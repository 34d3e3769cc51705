"""ShardedBatchedSocialNavGym with REAL environments: two ranks (gloo rendezvous on 127.0.0.1, both on GPU 0 -- a rehearsal of one
rank per GPU on a one-GPU box) step their shards of a 37-world hybrid batch generated on the device; the gathered observations,
rewards and end flags equal the single-process batch world for world, bit for bit.  The step has no collective."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOTAL, N = 37, 6


def _config():
    import configparser

    cfg = configparser.RawConfigParser()
    cfg.read_dict({
        "env": {"time_limit": 50, "time_step": 0.0125, "robot_time_step": 0.25, "val_size": 100, "test_size": 100, "randomize_attributes": "false"},
        "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2, "discomfort_penalty_factor": 0.5},
        "sim": {"train_val_sim": "hybrid_scenario", "test_sim": "hybrid_scenario", "square_width": 10, "circle_radius": 7, "human_num": N,
                "traffic_length": 14, "traffic_height": 3},
        "humans": {"visible": "true", "policy": "hsfm_farina", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
        "robot": {"visible": "false", "policy": "none", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
    })
    return cfg


def _actions(step):
    a = np.linspace(0, 2 * np.pi, TOTAL, endpoint=False) + 0.3 * step
    return (0.6 * np.stack([np.cos(a), np.sin(a)], -1)).astype(np.float32)


def _rollout(env, gather):
    out = [gather(env.reset(phase="val", first_case=5, device=True))]
    for k in range(6):
        obs, rew, term, trunc, info = env.step(_actions(k))
        out.append((gather(obs), gather(rew), gather(term), gather(trunc), gather(info)))
    return out


def _worker(rank, world_size, port, q):
    import torch.distributed as dist

    from social_navigation_pyenvs_amd.social_gym.sharded_gym import ShardedBatchedSocialNavGym

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    env = ShardedBatchedSocialNavGym(_config(), TOTAL, dist=dist)
    res = _rollout(env, env.gather)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, (env.first, env.W), res))


def test_two_ranks_equal_the_single_process_batch():
    import torch.multiprocessing as mp

    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == (0, 19) and res[1][1] == (19, 18)
    ref = _rollout(BatchedSocialNavGym(_config(), TOTAL), lambda x: x)
    for rank, _, got in res:
        np.testing.assert_array_equal(got[0], ref[0])
        for k in range(1, len(ref)):
            for a, b in zip(got[k], ref[k]):
                np.testing.assert_array_equal(a, b)
    assert np.any(ref[-1][0] != ref[0])


def test_shards_stay_the_single_process_batch_across_auto_resets():
    """Past the episode ends: a finished world's seed moves on by the worlds of the WHOLE job (cs_gym_book.seed_stride), so the ragged
    shards of a 37-world batch (19 + 18 worlds, stepped here one after the other on one GPU -- no communication is involved) keep
    returning what the single-process batch returns, world for world, through every world's second and third episode.  With the local
    shard size as the stride, rank 0's second episodes would be rank 1's first ones."""
    import torch

    from social_navigation_pyenvs_amd.social_gym.sharded_gym import ShardedBatchedSocialNavGym
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    full = BatchedSocialNavGym(_config(), TOTAL)
    full.reset(phase="val", first_case=5, device=True)
    shards = [ShardedBatchedSocialNavGym(_config(), TOTAL, rank=r, world_size=2) for r in range(2)]
    for s in shards:
        s.reset(phase="val", first_case=5, device=True)
    assert [(s.first, s.W) for s in shards] == [(0, 19), (19, 18)] and all(s.env.seed_stride == TOTAL for s in shards)
    seeds0 = full._device_loop_state()["seeds"].cpu().numpy().copy()
    ended = np.zeros(TOTAL, int)
    for k in range(90):
        rb = full.cw.d_robot.torch().view(TOTAL, 13)
        to_goal = rb[:, 10:12] - rb[:, 0:2]
        a = (to_goal / to_goal.norm(dim=1, keepdim=True).clamp(min=1e-6)).contiguous().clone()   # ReachGoal ends the episodes
        ref = [t.clone() for t in full.step_device(a)]
        got = [[t.clone() for t in s.step_device(a)] for s in shards]                     # every shard slices its rows out of the full array
        for j, r in enumerate(ref):
            assert torch.equal(torch.cat([g[j] for g in got], 0), r), (k, j)
        ended += (ref[2] | ref[3]).cpu().numpy().astype(int)
    assert (ended >= 2).all(), ended
    seeds = np.concatenate([s._dl["seeds"].cpu().numpy() for s in shards])
    np.testing.assert_array_equal(seeds, full._dl["seeds"].cpu().numpy())
    np.testing.assert_array_equal(seeds, seeds0 + TOTAL * ended)                       # world w walks s + w + k * total_worlds
    np.testing.assert_array_equal(np.concatenate([s.cw.get_states() for s in shards]), full.cw.get_states())
