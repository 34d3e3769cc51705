#!/usr/bin/env python3
"""Diagnostic: time of one batched imitation_learning_step (20 x { cs_robot_model_step ; cs_step(1 substep) }) at the bench
workload (4096 worlds x 25 HSFM humans, hybrid scenario), eager launches vs one HIP graph of the 40 launches."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: shares its HIP runtime)

from social_navigation_pyenvs_amd import _lib, scenarios as sc  # noqa: E402
from social_navigation_pyenvs_amd.batched import CrowdWorlds  # noqa: E402

W, n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 25
hmodel = sys.argv[2] if len(sys.argv) > 2 else "hsfm_farina"
for rmodel in (sys.argv[3:] or ["sfm_helbing", "hsfm_new_guo", "orca"]):
    S, goals, P, rb = sc.hybrid_worlds(W, n, hmodel)
    robot = np.zeros((W, 13), np.float32)
    robot[:, 0:2] = [0.0, -9.0]   # outside the crossing circle: no human spawns on top of it
    robot[:, 2] = np.pi / 2
    robot[:, 8], robot[:, 9], robot[:, 12] = 0.3, 80.0, 1.0
    robot[:, 10:12] = [0.0, 9.0]
    cw = CrowdWorlds(S, goals, P, None, None, type=hmodel, all_params_equal=True, respawn_bounds=rb,
                     respawn_worlds=(np.arange(W) % 2 == 1).astype(np.int32), layout="soa", robot=robot)
    cw.set_robot_model(rmodel, None if rmodel == "orca" else sc.default_params(rmodel), 0.01, np.full((W, n), 0.01, np.float32))
    stream = _lib.stream_create()
    cw.stream = stream
    for _ in range(3):
        cw.imitation_block(0.0125, 20)
    _lib.stream_sync(stream)
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        cw.imitation_block(0.0125, 20)
    _lib.stream_sync(stream)
    eager = (time.perf_counter() - t0) / K
    with _lib.Graph.capture(stream) as graph:
        for _ in range(K):
            cw.imitation_block(0.0125, 20, graph=False)
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record(stream)
    graph.launch()
    e1.record(stream)
    _lib.stream_sync(stream)
    g = e0.elapsed_ms(e1) / K
    rbx = cw.get_robot()
    print(f"W={W} humans={hmodel} robot={rmodel}: imitation step cs_imitation_block {eager * 1e6:.0f} us, the 40 alternating launches replayed from a HIP graph {g * 1e3:.0f} us "
          f"({W / (g * 1e-3):.3g} Gym steps/s); robots moved {np.mean(np.linalg.norm(rbx[:, 0:2] - robot[:, 0:2], axis=1)):.2f} m", flush=True)
