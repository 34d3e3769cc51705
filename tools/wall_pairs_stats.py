"""Diagnostic: cfg5's shard -- how many (agent, polygon) pairs are within the wall force's reach (+ a launch's travel) per world, and how
often some agent touches a polygon, along the bench window (Gym steps 20..70)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from social_navigation_pyenvs_amd import generators as gen, scenarios as sc

W, n = 2048, 50
cw = gen.static_obstacle_crossing(W, n, "hsfm_farina", first_world=0, radius=14.0, n_static=3, walls=True, layout="soa")
walls = np.asarray(sc.polygon_walls(), dtype=np.float64)      # [O][Smax][2][2]
O = walls.shape[0]
def seg_dist(p, seg):     # p [..., 2], seg [2][2]
    a, b = seg[0], seg[1]
    e = b - a
    t = np.clip(((p - a) @ e) / (e @ e), 0.0, 1.0)
    return np.linalg.norm(p - (a + t[..., None] * e), axis=-1)
B = 0.08; reach0 = 36.0 * B
for step in range(0, 71):
    if step in (0, 10, 20, 30, 40, 50, 60, 70):
        S = cw.get_states().astype(np.float64)
        p = S[:, :, 0:2]; r = S[:, :, 8]; vd = S[:, :, 12]
        pairs = np.zeros(W, int); pairs_exact = np.zeros(W, int); touch = np.zeros(W, bool); near_any = np.zeros((W, O), bool)
        for o in range(O):
            d = np.min(np.stack([seg_dist(p, walls[o, s]) for s in range(walls.shape[1]) if not np.isnan(walls[o, s]).any()]), axis=0)
            cen = np.nanmean(walls[o].reshape(-1, 2), axis=0); rad = np.nanmax(np.linalg.norm(walls[o].reshape(-1, 2) - cen, axis=1))
            dc = np.linalg.norm(p - cen, axis=-1)
            inreach = dc < rad + r + 0.01 + reach0 + 20 * 0.0125 * vd + 1e-3          # the kernel's bounding-circle test
            exact = (d - (r + 0.01)) < reach0 + 20 * 0.0125 * vd + 1e-3                        # ... and the distance to the polygon itself
            pairs_exact += exact.sum(1)
            pairs += inreach.sum(1); touch |= ((r + 0.01 - d) > -1e-3).any(1); near_any[:, o] = (dc < rad + r + 0.01 + reach0).any(1)
        print(f"Gym step {step:2d}: pairs per world mean {pairs.mean():5.1f} p90 {np.percentile(pairs, 90):4.0f} max {pairs.max():3d}; worlds with > 64 pairs {np.mean(pairs > 64):.3f}; by the exact distance: mean {pairs_exact.mean():5.1f} p99 {np.percentile(pairs_exact, 99):4.0f} max {pairs_exact.max():3d}; "
              f"worlds with a contact now {touch.mean():.3f}; polygons somebody is near: {near_any.mean():.2f}", flush=True)
    cw.step(0.0125, 20)
