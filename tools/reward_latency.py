#!/usr/bin/env python3
"""Time of cs_collision_reward on the bench batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from social_navigation_pyenvs_amd import _lib, scenarios as sc
from social_navigation_pyenvs_amd.batched import CrowdWorlds
import ctypes as C
for W, n, layout in ((4096, 25, "soa"), (4096, 25, "aos"), (16384, 25, "soa"), (8192, 50, "soa")):
    S, goals, P, rb = sc.hybrid_worlds(W, n, "hsfm_farina")
    robot = np.zeros((W, 13), np.float32); robot[:, 8] = 0.3; robot[:, 10:12] = (0, 7)
    cw = CrowdWorlds(S, goals, P, None, None, type="hsfm_farina", all_params_equal=True, robot=robot, layout=layout)
    act = np.zeros((W, 2), np.float32); gt = np.zeros(W, np.float32)
    cw.collision_reward(act, 0.25, gt)
    e0, e1 = _lib.Event(), _lib.Event()
    d = cw.descriptor(); a = cw._upload("reward_action", act); g = cw._upload("global_time", gt); out = cw._buffer("reward_out", (W, 7))
    cfg = (C.c_float * 5)(50.0, 1.0, -0.25, 0.2, 0.5)
    e0.record(cw.stream)
    for _ in range(50):
        _lib.check(_lib.load().cs_collision_reward(C.byref(d), C.c_void_p(a.ptr), C.c_float(0.25), C.c_void_p(g.ptr), cfg, C.c_void_p(out.ptr), C.c_void_p(cw.stream)))
    e1.record(cw.stream)
    print(f"cs_collision_reward W={W} n={n} {layout}: {e0.elapsed_ms(e1) / 50 * 1e3:.1f} us")
