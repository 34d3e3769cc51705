#!/usr/bin/env python3
"""PCIe-inclusive rate of the array seam (host numpy in / out every call) vs the resident path."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p_)
from social_navigation_pyenvs_amd import scenarios as sc
from social_navigation_pyenvs_amd.social_gym.src.forces_parallel import update_humans_parallel

W, n = 4096, 25
S, goals, P, rb = sc.hybrid_worlds(W, n, "hsfm_farina")
saf = np.zeros((W, n))
for _ in range(3):
    out = update_humans_parallel(3, S, goals, None, P, 0.0125, saf, True, False)
t0 = time.perf_counter(); reps = 20
for _ in range(reps):
    out = update_humans_parallel(3, S, goals, None, P, 0.0125, saf, True, False)
el = (time.perf_counter() - t0) / reps
print(f"array seam W={W} N={n}: {el*1e3:.3f} ms per substep call incl. H2D/D2H/alloc -> {W*n/el/1e6:.1f} M agent-substeps/s")
# W=1 facade Gym step
from test_facade_cpu import make_env
from social_navigation_pyenvs_amd.crowd_nav.utils.action import ActionXY
env = make_env("hsfm_farina", "circle_crossing", 25, False)
env.reset(phase="test", test_case=1)
for _ in range(3): env.step(ActionXY(0.1, 0.5))
t0 = time.perf_counter()
for _ in range(50): env.step(ActionXY(0.1, 0.5))
el = (time.perf_counter() - t0) / 50
print(f"W=1 SocialNavGym.step (25 humans, 20 fused substeps, host objects refreshed): {el*1e3:.3f} ms per Gym step")
