"""Diagnostic: dump every agent-substep on which a fast ORCA arithmetic (cs_worlds.orca_math fast / fma) is beyond 1e-5 of the exact restatement and
the double evaluation does not explain it (classes edge1 / edge4 / edge16 / unexplained of tests/orca_fast_parity.py), with the world's input rows,
so the cases can be classified offline on the CPU (tools/orca_classify.py).
  python tools/orca_unexplained_dump.py [out.npz]   -> gpurun_out/orca_unexplained.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402

import orca_fast_parity as ofp  # noqa: E402
from social_navigation_pyenvs_amd import _lib  # noqa: E402
from social_navigation_pyenvs_amd.batched import CrowdWorlds  # noqa: E402

OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "orca_unexplained.npz")
lib = _lib.load()
recs = []
# the shapes and seeds of tests/test_gpu_orca_fast.py
for mode, name in ((2, "fma"), (1, "fast")):
    for W, n, R, nsub in ((512, 25, 7.0, 700), (256, 10, 3.0, 300), (128, 40, 6.0, 300)):
        S, g, margin = ofp.crossing(W, n, R, 4242 + n)
        cw = CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa", orca_math=name)
        got = []
        res = ofp.substeps_vs_restatement(cw, S, g, margin, 0.0125, nsub, seed=mode, collect=got, progress=50)
        for r in got:
            r.update(mode=mode, n=n)
        recs += got
        print(f"{name} {W}x{n}: beyond {res['beyond_bar']} f64 {res['class_f64']} e1 {res['class_edge1']} e4 {res['class_edge4']} e16 {res['class_edge16']} "
              f"op4 {res['class_op4']} unexplained {res['unexplained']} decisions {res['decisions']} worst {res['worst_unexplained']:.3g}; collected {len(got)}", flush=True)
out = {}
for i, r in enumerate(recs):
    for k, v in r.items():
        out[f"{i}_{k}"] = np.asarray(v)
out["count"] = np.asarray(len(recs))
os.makedirs(os.path.dirname(OUT), exist_ok=True)
np.savez_compressed(OUT, **out)
print("written", OUT, len(recs))
