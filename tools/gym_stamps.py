#!/usr/bin/env python3
"""usage (GPU box): tools/gym_stamps.py [worlds=4096] [humans=25]
The diagnostic build's stamps (tools/stamp_probe.py) inside the DEVICE-RESIDENT GYM STEP instead of the bare cs_step: what the head (swept
collision test, reward, bookkeeping), the observation and the take-over logic add to a launch, per phase -- entry -> loads returned, -> first
substep (the head runs here), the substep loop, last substep -> stores issued (observation, take-over) -- beside the same figures of the bare
step on the same worlds."""
import configparser
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from social_navigation_pyenvs_amd import _lib  # noqa: E402
from social_navigation_pyenvs_amd.csrc import build as hb  # noqa: E402

so = hb.build(variant="stamps", extra_flags=["-DCS_STAMPS"])
_lib.LIB_PATH = so
_lib._lib = None
import torch  # noqa: E402
from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
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
cw = env.cw
g, b, wpb = cw.launch_geometry()
buf = _lib.DeviceBuffer((g * (b // 64), 20), np.uint64)
_lib.load().cs_debug_set_stamp_buffer(C.c_void_p(buf.ptr))
act = env.action_buffer()
act.zero_()
names = ["entry -> every load returned", "-> first substep (prologue; Gym: the head)", "the 20 substeps", "last substep -> stores issued (Gym: observation, take-over)"]


def collect(step, reps=12):
    acc = np.zeros((reps, 4))
    life = np.zeros(reps)
    for k in range(reps):
        step()
        torch.cuda.synchronize()
        st = buf.download().astype(np.float64)
        acc[k] = [st[:, 12].mean(), st[:, 13].mean(), st[:, :11].sum(1).mean(), st[:, 14].mean()]
        life[k] = 10e-3 * (st[:, 18] - st[:, 17]).mean()
    return np.median(acc, axis=0), float(np.median(life))


with torch.cuda.stream(env.device_stream()):
    for _ in range(20):
        env.step_device(act, auto_reset=True)
    torch.cuda.synchronize()
    for label, mode in (("Gym step, same-step auto-reset", True), ("Gym step, no resets", False)):
        v, life = collect(lambda: env.step_device(act, auto_reset=mode))
        print(f"{label}: a wavefront lives {life:.2f} us (wall clock)")
        for nm, x in zip(names, v):
            print(f"    {nm:62s} {x:9.0f} shader clocks")
    v, life = collect(lambda: (cw.step(0.0125, 20), cw.sync()))
    print(f"bare cs_step on the same worlds: a wavefront lives {life:.2f} us (wall clock)")
    for nm, x in zip(names, v):
        print(f"    {nm:62s} {x:9.0f} shader clocks")
env.close()
