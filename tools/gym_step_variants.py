#!/usr/bin/env python3
"""us per batched Gym step (same-step auto-reset, from inside the library's stream) for a few refill cadences / staging depths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym
import configparser

W, n = 4096, 25
cfg = configparser.RawConfigParser()
cfg.read_dict({
    "env": {"time_limit": 50, "time_step": 0.0125, "robot_time_step": 0.25, "val_size": 100, "test_size": 500, "randomize_attributes": "false"},
    "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2, "discomfort_penalty_factor": 0.5},
    "sim": {"train_val_sim": "hybrid_scenario", "test_sim": "hybrid_scenario", "square_width": 10, "circle_radius": 7, "human_num": n, "traffic_length": 14, "traffic_height": 3},
    "humans": {"visible": "true", "policy": "hsfm_farina", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
    "robot": {"visible": "false", "policy": "none", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
})
VARIANTS = [(64, 64, True, 1), (64, 64, True, 0), (32, 64, True, 1), (64, 128, True, 1), (128, 128, True, 1), (128, 128, True, 0), (64, 64, "next_step", 1), (64, 64, "next_step", 0), (10 ** 9, 16, False, 1)]
for every, depth, mode, prio in VARIANTS:
    env = BatchedSocialNavGym(cfg, W)
    env.REFILL_EVERY, env.STAGE_DEPTH, env.REFILL_PRIORITY = every, depth, prio
    env.reset(phase="train", first_case=0, device=True)
    buf = env.action_buffer()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    buf.copy_(torch.randn(W, 2, device="cuda", generator=g) * 0.5)
    with torch.cuda.stream(env.device_stream()):
        for _ in range(60):
            env.step_device(buf, auto_reset=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            env.step_device(buf, auto_reset=mode)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 400 * 1e6
    dl = env._dl
    stale = int((dl["epoch"] > 0).sum())
    print(f"prio {prio} refill every {every:>10d} depth {depth:3d} mode {str(mode):9s}: {el:6.1f} us per step; worlds that ended {stale}, max epoch {int(dl['epoch'].max())}, failed {env.failed_resets()}")
    del env
