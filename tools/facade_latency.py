"""What ONE env.step() of the drop-in single-environment facade costs (SocialNavGym.step, /root/reference/social_gym/social_nav_gym.py:227-250;
BASELINE.md quotes 227 ms per reference Gym step at N = 25): microseconds per call at W = 1 for N = 5 / 25 hsfm_farina humans, with the SARL-style
peek (MotionModelManager.get_next_human_observable_states, motion_model_manager.py:691-709) a policy issues before every step, split into
  build    MotionModelManager._device(): host mirrors -> device (buffers, uploads)
  launch   the fused 20-substep launch (cs_step) / the peek launch, until the stream is idle
  readback device -> host mirrors and agent objects
  host     everything else in step(): the swept collision test and reward in Python, state records, the observation list
  python tools/facade_latency.py [steps=300] [out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from test_facade_cpu import make_env  # noqa: E402  (the facade wired like the reference's tests wire theirs)

from social_navigation_pyenvs_amd import _lib  # noqa: E402
from social_navigation_pyenvs_amd.crowd_nav.utils.action import ActionXY  # noqa: E402
from social_navigation_pyenvs_amd.social_gym.src import motion_model_manager as mmm  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "facade_latency.json")
acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return wrapper


M = mmm.MotionModelManager
M._device = timed("build", M._device)
M._readback = timed("readback", M._readback)
from social_navigation_pyenvs_amd import batched  # noqa: E402


def _sync_after(fn):
    def wrapper(self, *a, **k):
        out = fn(self, *a, **k)
        self.sync()
        return out
    return wrapper


batched.CrowdWorlds.step = timed("launch", _sync_after(batched.CrowdWorlds.step))
batched.CrowdWorlds.peek = timed("peek_launch_and_download", batched.CrowdWorlds.peek)

report = {"device": _lib.device_name(0), "steps": STEPS, "reference_ms_per_gym_step_N25": 227.0, "rows": []}
for n, visible in ((5, False), (25, False), (25, True)):
    env = make_env("hsfm_farina", "circle_crossing", n, visible)
    env.reset(phase="test", test_case=1)
    rng = np.random.default_rng(n)
    for warm in range(20):
        env.step(ActionXY(0.1, 0.1))
    acc.clear()
    t_peek = t_step = 0.0
    for k in range(STEPS):
        if k % 60 == 0:
            env.reset(phase="test", test_case=1 + k // 60)
        a = rng.uniform(-0.5, 0.5, 2)
        t0 = time.perf_counter()
        env.motion_model_manager.get_next_human_observable_states(env.robot_time_step)
        t1 = time.perf_counter()
        env.step(ActionXY(float(a[0]), float(a[1])))
        t2 = time.perf_counter()
        t_peek += t1 - t0; t_step += t2 - t1
    us = lambda x: round(x / STEPS * 1e6, 1)
    row = {"humans": n, "robot_visible": visible, "step_us": us(t_step), "peek_us": us(t_peek), "step_plus_peek_us": us(t_step + t_peek),
           "build_us": us(acc.get("build", 0.0)), "launch_us": us(acc.get("launch", 0.0)), "readback_us": us(acc.get("readback", 0.0)),
           "peek_launch_and_download_us": us(acc.get("peek_launch_and_download", 0.0)),
           "reference_over_this": round(227.0e3 / us(t_step), 1) if n == 25 else None}
    row["host_python_us"] = round(row["step_plus_peek_us"] - row["build_us"] - row["launch_us"] - row["readback_us"] - row["peek_launch_and_download_us"], 1)
    report["rows"].append(row)
    print(row, flush=True)
os.makedirs(os.path.dirname(OUT), exist_ok=True)
json.dump(report, open(OUT, "w"), indent=1)
print("written", OUT)
