#!/usr/bin/env python3
"""usage (GPU box): tools/gym_step_floor.py [worlds=4096] [humans=25]
The device-resident Gym step against the number of fused substeps (robot_time_step / time_step = 1, 5, 10, 20, 40), without resets: the
slope is a substep, the intercept what a Gym step costs around its substeps (launch, loads, the head: swept collision test + reward +
bookkeeping, the observation, the stores).  Beside it the bare cs_step of the same worlds (a loop of plain launches on the same stream)."""
import configparser
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
steps = 200
rows = []
for nsub in (1, 5, 10, 20, 40):
    cfg = configparser.RawConfigParser()
    cfg.read_dict({
        "env": {"time_limit": 5000, "time_step": 0.0125, "robot_time_step": 0.0125 * nsub, "val_size": 100, "test_size": 500, "randomize_attributes": "false"},
        "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2, "discomfort_penalty_factor": 0.5},
        "sim": {"train_val_sim": "hybrid_scenario", "test_sim": "hybrid_scenario", "square_width": 10, "circle_radius": 7, "human_num": n,
                "traffic_length": 14, "traffic_height": 3},
        "humans": {"visible": "true", "policy": "hsfm_farina", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
        "robot": {"visible": "false", "policy": "none", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
    })
    env = BatchedSocialNavGym(cfg, W)
    env.reset(phase="train", first_case=0, device=True)
    buf = env.action_buffer()
    buf.zero_()
    # a crowd's cost follows its state (contacts, respawns): the two loops alternate in blocks of 25 steps over the same stretch of the episode
    with torch.cuda.stream(env.device_stream()):
        cw = env.cw
        for _ in range(30):
            env.step_device(buf, auto_reset=False)
        torch.cuda.synchronize()
        gym_t = bare_t = 0.0
        for _ in range(steps // 25):
            t0 = time.perf_counter()
            for _ in range(25):
                env.step_device(buf, auto_reset=False)
            torch.cuda.synchronize()
            gym_t += time.perf_counter() - t0
            t0 = time.perf_counter()
            for _ in range(25):
                cw.step(0.0125, nsub)
            cw.sync()
            torch.cuda.synchronize()
            bare_t += time.perf_counter() - t0
        gym_us, bare_us = gym_t / (steps // 25 * 25) * 1e6, bare_t / (steps // 25 * 25) * 1e6
    rows.append((nsub, gym_us, bare_us))
    print(f"substeps {nsub:3d} | Gym step {gym_us:7.2f} us | bare cs_step (plain launches) {bare_us:7.2f} us | difference {gym_us - bare_us:6.2f} us", flush=True)
    env.close()
a = np.array(rows)
for k, nm in ((1, "Gym step"), (2, "bare cs_step")):
    slope, icpt = np.polyfit(a[:, 0], a[:, k], 1)
    print(f"{nm}: {slope:.3f} us per substep + {icpt:.2f} us per launch")
