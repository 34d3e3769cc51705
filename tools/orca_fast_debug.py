"""Diagnostic: find agent-substeps where the fast ORCA arithmetic is far (> thr) from the exact restatement and dump their worlds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import orca_fast_parity as ofp
from oracle import crowd_oracle as orc
from social_navigation_pyenvs_amd import _lib
from social_navigation_pyenvs_amd.batched import CrowdWorlds
W, n, R, mode, thr = 4096, 25, 7.0, int(sys.argv[1]) if len(sys.argv) > 1 else 1, float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
S, g, margin = ofp.crossing(W, n, R, 31337 + n)
cw = CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa", orca_math=("exact", "fast", "fma")[mode])
ref, rg = S.copy(), g.copy()
dump = []
for k in range(700):
    cw.set_states(ref); cw.set_goals(rg); cw.step(0.0125, 1)
    got = cw.get_states()
    nxt, ng, _ = orc.orca_step_block(ref, rg, margin, 0.0125, 1)
    err = np.abs(got[..., [3, 4]].astype(np.float64) - nxt[..., [3, 4]]).max(axis=-1)
    for w_, a_ in zip(*np.nonzero(err > thr)):
        if len(dump) < 40:
            dump.append(dict(k=k, w=int(w_), a=int(a_), S=ref[w_].copy(), g=rg[w_].copy(), got=got[w_].copy(), nxt=nxt[w_].copy()))
            print(k, w_, a_, "err", err[w_, a_], "got v", got[w_, a_, 3:5], "ref v", nxt[w_, a_, 3:5], flush=True)
    ref, rg = nxt, ng
np.savez(os.path.join(ROOT, "gpurun_out", f"orca_fast_debug_m{mode}.npz"), **{f"{i}_{k}": v for i, d in enumerate(dump) for k, v in d.items()})
