"""Diagnostic: how much of a cfg4 launch is imbalance between wavefronts?  The same 4096 worlds of the dense phase are stepped (one launch of 20
substeps, restored each time) in three world orders: as generated, heavy worlds paired with light ones in a wavefront (worlds sorted by a
crowding proxy, first with last), and heavy with heavy (sorted order).  The worlds and their results are the same; only which two share a
wavefront changes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import orca_fast_parity as ofp
from social_navigation_pyenvs_amd import _lib
from social_navigation_pyenvs_amd.batched import CrowdWorlds

W, n = 4096, 25
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 30
S, g, margin = ofp.crossing(W, n, 7.0, 1000)
cw = CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa")
for _ in range(warm):
    cw.step(0.0125, 20)
S1, g1 = cw.get_states(), cw.get_goals()
dd = np.linalg.norm(S1[:, :, None, 0:2] - S1[:, None, :, 0:2], axis=-1) + 10 * np.eye(n)[None]
cost = (dd < 1.2).sum(axis=(1, 2)).astype(np.float64) + 1e-3 * (dd < 2.0).sum(axis=(1, 2))
order = np.argsort(cost)
paired = np.empty(W, dtype=np.int64); paired[0::2] = order[: W // 2]; paired[1::2] = order[::-1][: W // 2]
orders = {"as generated": np.arange(W), "heavy + light per wavefront": paired, "heavy + heavy (sorted)": order, "random": np.random.default_rng(0).permutation(W)}
ev0, ev1 = _lib.Event(), _lib.Event()
for name, perm in orders.items():
    c = CrowdWorlds(S1[perm], g1[perm], None, margin[perm], None, type="orca", layout="soa")
    best = []
    for rep in range(7):
        c.set_states(S1[perm]); c.set_goals(g1[perm]); c.sync()
        ev0.record(c.stream); c.step(0.0125, 20); ev1.record(c.stream); c.sync()
        best.append(ev0.elapsed_ms(ev1) * 1e3)
    print(f"{name:32s} {np.median(best):8.1f} us (min {min(best):.1f})  default math={_lib.load().cs_orca_default_math()}", flush=True)
print("crowding proxy: min %.0f median %.0f max %.0f" % (cost.min(), np.median(cost), cost.max()))
