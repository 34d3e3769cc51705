#!/usr/bin/env python3
"""Host-side cost of the pieces of BatchedSocialNavGym._maybe_refill inside a running device loop (us per call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from social_navigation_pyenvs_amd import _lib
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
env = BatchedSocialNavGym(cfg, W)
env.reset(phase="train", first_case=0, device=True)
buf = env.action_buffer()
buf.copy_(torch.randn(W, 2, device="cuda") * 0.5)
dl = env._device_loop_state()
lib = _lib.load()
T = {"query": [], "launch": [], "record": [], "graph": []}
with torch.cuda.stream(env.device_stream()):
    for _ in range(50):
        env.step_device(buf)
    torch.cuda.synchronize()
    env.REFILL_EVERY = 10 ** 9       # by hand below
    t_all = time.perf_counter()
    for k in range(400):
        t0 = time.perf_counter(); env.step_device(buf); t1 = time.perf_counter()
        T["graph"].append(t1 - t0)
        if k % 4 == 3:
            t0 = time.perf_counter(); d = dl["refill_ev"].done(); t1 = time.perf_counter()
            T["query"].append(t1 - t0)
            if d:
                t0 = time.perf_counter(); _lib.check(lib.cs_refill_staged_worlds(*dl["refill_args"])); t1 = time.perf_counter()
                dl["refill_ev"].record(dl["stream_b"]); t2 = time.perf_counter()
                T["launch"].append(t1 - t0); T["record"].append(t2 - t1)
    torch.cuda.synchronize()
    print("per step %.1f us" % ((time.perf_counter() - t_all) / 400 * 1e6))
import numpy as np
for k, v in T.items():
    v = np.array(v) * 1e6
    print(f"{k:8s} n {len(v):4d} mean {v.mean():7.1f} median {np.median(v):7.1f} p90 {np.percentile(v, 90):7.1f} us")
