#!/usr/bin/env python3
"""End-to-end rate of the device-resident Gym loop: BatchedSocialNavGym.step_device = swept collision + reward,
20 fused substeps, masked device reset of finished worlds, observation gather -- actions and observations stay in HBM."""
import configparser
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cfg = configparser.RawConfigParser()
cfg.read_dict({
    "env": {"time_limit": 50, "time_step": 0.0125, "robot_time_step": 0.25, "val_size": 100, "test_size": 500, "randomize_attributes": "false"},
    "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2, "discomfort_penalty_factor": 0.5},
    "sim": {"train_val_sim": "hybrid_scenario", "test_sim": "hybrid_scenario", "square_width": 10, "circle_radius": 7, "human_num": n,
            "traffic_length": 14, "traffic_height": 3},
    "humans": {"visible": "true", "policy": "hsfm_farina", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
    "robot": {"visible": "false", "policy": "none", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
})
env = BatchedSocialNavGym(cfg, W)
t0 = time.perf_counter()
env.reset(phase="train", first_case=0, device=True)
torch.cuda.synchronize()
print(f"device reset of {W} x {n}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
for _ in range(20):
    env.step_device(torch.randn(W, 2, device="cuda") * 0.5)
torch.cuda.synchronize()
K = 400
ended = 0
t0 = time.perf_counter()
for _ in range(K):
    ob, rew, term, trunc, info = env.step_device(torch.randn(W, 2, device="cuda") * 0.5)
    ended += (term | trunc).sum()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{K} Gym steps of {W} worlds x {n} humans: {el / K * 1e6:.1f} us per batched step, {W * K / el:.3e} Gym steps/s, "
      f"{W * K * n * 20 / el:.3e} agent-substeps/s, {int(ended)} episodes ended and were regenerated on the device")

# the loop alone: actions written into the persistent buffer once, nothing else launched per step
buf = env.action_buffer()
buf.copy_(torch.randn(W, 2, device="cuda") * 0.5)
for _ in range(20):
    env.step_device(buf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    env.step_device(buf)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"graph replay only (actions in place, no per-step torch ops): {el / K * 1e6:.1f} us per batched step, {W * K / el:.3e} Gym steps/s")
for _ in range(20):
    env.step_device(buf, auto_reset=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    env.step_device(buf, auto_reset=False)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"graph replay without the auto-reset branch: {el / K * 1e6:.1f} us per batched step")
with torch.cuda.stream(env.device_stream()):       # the caller's own work on the library's stream: no cross-stream waits in step_device
    for _ in range(20):
        env.step_device(buf, auto_reset=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        env.step_device(buf, auto_reset=False)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"   the same from inside `with torch.cuda.stream(env.device_stream())`: {el / K * 1e6:.1f} us per batched step")
    t0 = time.perf_counter()
    for _ in range(K):
        env.step_device(buf)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"   ... with same-step auto-reset: {el / K * 1e6:.1f} us per batched step")

env2 = BatchedSocialNavGym(cfg, W)
env2.reset(phase="train", first_case=0, device=True)
buf2 = env2.action_buffer()
buf2.copy_(torch.randn(W, 2, device="cuda") * 0.5)
for _ in range(20):
    env2.step_device(buf2, auto_reset="next_step")
torch.cuda.synchronize()
t0 = time.perf_counter()
ended2 = torch.zeros((), dtype=torch.int64, device="cuda")
for _ in range(K):
    ob, rew, term, trunc, info = env2.step_device(buf2, auto_reset="next_step")
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"NEXT_STEP autoreset (regeneration beside the following step): {el / K * 1e6:.1f} us per batched step, {W * K / el:.3e} Gym steps/s")
t0 = time.perf_counter()
for _ in range(K):
    env2.step_device(buf2, auto_reset="next_step")
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"   of which host-side launch time {t_host / K * 1e6:.1f} us per step (total {el / K * 1e6:.1f})")
with torch.cuda.stream(env2.device_stream()):
    for _ in range(20):
        env2.step_device(buf2, auto_reset="next_step")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        env2.step_device(buf2, auto_reset="next_step")
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"   NEXT_STEP from inside `with torch.cuda.stream(env.device_stream())`: {el / K * 1e6:.1f} us per batched step, {W * K / el:.3e} Gym steps/s")

