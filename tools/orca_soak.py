"""Diagnostic: the ORCA kernel (register-resident LP2 / LP3) against the C restatement, bit for bit, on many worlds and substeps."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch  # noqa: F401
from oracle import crowd_oracle as orc
from social_navigation_pyenvs_amd import scenarios as sc
from social_navigation_pyenvs_amd.batched import CrowdWorlds

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for n, R in ((25, 7.0), (10, 3.0), (40, 6.0)):
    pos, yaw, g = sc.circular_crossing(W, n, R, 31337 + n)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    d = g[:, :, 0] - S[:, :, 0:2]
    S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    margin = np.full((W, n), 0.01, np.float32)
    cw = CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa")
    ref, rg = S, g
    t0 = time.time()
    bad = 0
    for b in range(blocks):
        cw.step(0.0125, 20)
        ref, rg, _ = orc.orca_step_block(ref, rg, margin, 0.0125, 20)
        got = cw.get_states()
        bad += int(np.sum(np.any(got[..., [0, 1, 3, 4, 5, 6]] != ref[..., [0, 1, 3, 4, 5, 6]], axis=(1, 2))))
    moved = float(np.mean(np.linalg.norm(ref[..., 0:2] - S[..., 0:2], axis=-1)))
    print(f"ORCA n={n} R={R}: {W} worlds x {blocks * 20} substeps, worlds that ever differed: {bad}; mean displacement {moved:.2f} m  ({time.time() - t0:.0f} s)", flush=True)
