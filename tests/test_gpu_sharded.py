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
