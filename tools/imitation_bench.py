#!/usr/bin/env python3
"""Diagnostic: time of one batched imitation_learning_step's substep loop (cs_imitation_block: 20 x { update_robot ; update_humans })
at the bench workload (4096 worlds x 25 HSFM humans, hybrid scenario) for an INVISIBLE and a VISIBLE robot, each as the library's
fused form (invisible: crowd launch with snapshots + robot launch; visible: ONE launch, the robot as the last row of the crowd's
kernel) and as the reference's strict alternation of 40 launches -- both replayed from a HIP graph, timed with HIP events."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: shares its HIP runtime)

from social_navigation_pyenvs_amd import _lib, scenarios as sc  # noqa: E402
from social_navigation_pyenvs_amd.batched import CrowdWorlds  # noqa: E402

W, n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 25
hmodel = sys.argv[2] if len(sys.argv) > 2 else "hsfm_farina"
K = 20


def timed(cw, stream, fused):
    cw.reserve_scratch(20)
    with _lib.Graph.capture(stream) as graph:
        for _ in range(K):
            cw.imitation_block(0.0125, 20, graph=fused)
    graph.launch()
    _lib.stream_sync(stream)
    e0, e1 = _lib.Event(), _lib.Event()
    best = []
    for _ in range(5):
        e0.record(stream)
        graph.launch()
        e1.record(stream)
        _lib.stream_sync(stream)
        best.append(e0.elapsed_ms(e1) / K)
    return float(np.median(best)) * 1e3


for rmodel in (sys.argv[3:] or ["sfm_helbing", "hsfm_new_guo", "orca"]):
    for visible in (False, True):
        S, goals, P, rb = sc.hybrid_worlds(W, n, hmodel)
        robot = np.zeros((W, 13), np.float32)
        robot[:, 0:2] = [0.0, -9.0]   # outside the crossing circle: no human spawns on top of it
        robot[:, 2] = np.pi / 2
        robot[:, 8], robot[:, 9], robot[:, 12] = 0.3, 80.0, 1.0
        robot[:, 10:12] = [0.0, 9.0]
        St = np.concatenate([S, robot[:, None, :]], axis=1) if visible else S
        out = {}
        for fused in (True, False):
            stream = _lib.stream_create()
            cw = CrowdWorlds(St, goals, P, None, None, type=hmodel, all_params_equal=True, respawn_bounds=rb,
                             respawn_worlds=(np.arange(W) % 2 == 1).astype(np.int32), layout="soa", robot=robot, robot_row=visible, stream=stream)
            cw.set_robot_model(rmodel, None if rmodel == "orca" else sc.default_params(rmodel), 0.01, np.full((W, n + int(visible)), 0.01, np.float32))
            for _ in range(3):
                cw.imitation_block(0.0125, 20, graph=fused)
            out[fused] = timed(cw, stream, fused)
            moved = float(np.mean(np.linalg.norm(cw.get_robot()[:, 0:2] - robot[:, 0:2], axis=1)))
        print(f"W={W} humans={hmodel} robot={rmodel} {'VISIBLE' if visible else 'invisible'}: cs_imitation_block {out[True]:.0f} us per Gym step "
              f"({W / (out[True] * 1e-6):.3g} Gym steps/s); the 40 alternating launches {out[False]:.0f} us; robots moved {moved:.2f} m", flush=True)
