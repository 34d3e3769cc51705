#!/usr/bin/env python3
"""Latency of cs_generate_worlds for ONE world (what a masked auto-reset pays) and time of a full batch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from social_navigation_pyenvs_amd import _lib  # noqa: E402
from social_navigation_pyenvs_amd.batched import CrowdWorlds  # noqa: E402
from social_navigation_pyenvs_amd import generators as gen  # noqa: E402

for scenario, n, W in (("circle_crossing", 25, 1), ("parallel_traffic", 25, 1), ("hybrid_scenario", 25, 1),
                       ("circle_crossing", 5, 1), ("hybrid_scenario", 25, 4096), ("circle_crossing", 50, 8192)):
    G = 1 if scenario == "parallel_traffic" else 2
    cw = CrowdWorlds(np.zeros((W, n, 13)), np.full((W, n, G, 2), np.nan), np.zeros((n, 20)), None, None, type="sfm_helbing",
                     all_params_equal=True, robot=np.zeros((W, 13)), respawn_worlds=np.zeros(W, np.int32))
    g = gen.make_generator(cw, scenario, circle_radius=(20 if n == 50 else 7))
    seeds = cw._upload("s", (1000 + np.arange(W)).astype(np.uint32), np.uint32)
    gen.generate_worlds_device(cw, g, seeds)
    cw.sync()
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record(cw.stream)
    for _ in range(10):
        gen.generate_worlds_device(cw, g, seeds)
    e1.record(cw.stream)
    print(f"{scenario:40s} n={n:3d} W={W:5d}: {e0.elapsed_ms(e1) / 10 * 1e3:9.1f} us per launch")
