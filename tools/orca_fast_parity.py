"""Diagnostic / evidence: the ORCA kernel's arithmetic modes (exact / fast / fma, cs_worlds.orca_math) per substep against the exact
restatement, with every agent-substep beyond 1e-5 examined for a decision edge (tests/orca_fast_parity.py), the free-running
health of each build, and the kernel time per mode.
  python tools/orca_fast_parity.py [worlds=4096] [substeps=700] [out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (the HIP runtime torch ships)

import orca_fast_parity as ofp  # noqa: E402
from social_navigation_pyenvs_amd import _lib  # noqa: E402
from social_navigation_pyenvs_amd.batched import CrowdWorlds  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NSUB = int(sys.argv[2]) if len(sys.argv) > 2 else 700
OUT = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "orca_fast_parity.json")
MODES = {0: "exact", 1: "fast", 2: "fma"}
lib = _lib.load()
report = {"device": _lib.device_name(0), "bar": ofp.BAR, "protocol": "one substep from re-synchronised rows, vs oracle/orca_oracle.c (float32, exact); edge = the restatement itself moves >= bar under 1-ulp input noise", "shapes": {}}


def timed_launch(cw, dt, nsub, reps=5):
    ev0, ev1 = _lib.Event(), _lib.Event()
    best = 1e9
    snap = cw.get_states(); gs = cw.get_goals()
    for _ in range(reps):
        cw.set_states(snap); cw.set_goals(gs)
        ev0.record(cw.stream)
        cw.step(dt, nsub)
        ev1.record(cw.stream)
        cw.sync()
        best = min(best, ev0.elapsed_ms(ev1))
    cw.set_states(snap); cw.set_goals(gs)
    return best * 1e3


for n, R, Wn, nsub in ((25, 7.0, W, NSUB), (10, 3.0, max(256, W // 4), min(NSUB, 400)), (40, 6.0, max(256, W // 4), min(NSUB, 400))):
    S, g, margin = ofp.crossing(Wn, n, R, 31337 + n)
    key = f"{Wn}x{n}_R{R:g}"
    report["shapes"][key] = {}
    for mode in (0, 1, 2):
        cw = CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa", orca_math=MODES[mode])
        t0 = time.time()
        res = ofp.substeps_vs_restatement(cw, S, g, margin, 0.0125, nsub, progress=100 if mode else None, seed=mode)
        res["variant"] = cw.step_variant()
        res["seconds"] = round(time.time() - t0, 1)
        if mode == 0:
            assert res["bit_identical_agent_substeps"] == res["agent_substeps"], "the exact build must be bit-identical to the restatement"
        report["shapes"][key][MODES[mode]] = res
        print(f"[{key}] {MODES[mode]:5s}: {res['agent_substeps']:.3g} agent-substeps, bit-identical {res['bit_identical_agent_substeps'] / res['agent_substeps']:.4f}, "
              f"p50 {res['p50']:.1e} p99 {res['p99']:.1e} p99.99 {res['p9999']:.1e} worst {res['worst']:.2e}; beyond {ofp.BAR:g} vs exact f32: {res['beyond_bar']} "
              f"({res['beyond_bar_share']:.2e} of all) = f64 {res['class_f64']} + edge1 {res['class_edge1']} + edge4 {res['class_edge4']} + edge16 {res['class_edge16']} + op4 {res['class_op4']} + unexplained {res['unexplained']} "
              f"(worst {res['worst_unexplained']:.2e}) + not examined {res['not_examined']}; of these closer to f64: build {res['disagree_build_closer_to_f64']} / exact {res['disagree_exact_closer_to_f64']}; "
              f"beyond bar vs f64: build {res['beyond_bar_vs_f64_build_share']:.2e}, exact f32 {res['beyond_bar_vs_f64_exact_share']:.2e}; goal flips {res['goal_column_flips']}; "
              f"pref velocity worst {res['pref_velocity_worst']:.1e}; decisions {res['decisions']}; probe reproduces the build's answer on {res['probe_reproduces_build']}  ({res['seconds']} s)", flush=True)
    # free-running health and kernel time per mode (same worlds, same launch shape as bench.py's cfg4: 20 fused substeps)
    for mode in (0, 1, 2):
        h = ofp.free_run_health(lambda: CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa", orca_math=MODES[mode]), S, g, margin, 0.0125, nsub)
        cw = CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa", orca_math=MODES[mode])
        for _ in range(25):
            cw.step(0.0125, 20)          # into the dense phase
        h["dense_launch_us"] = round(timed_launch(cw, 0.0125, 20), 1)
        report["shapes"][key][MODES[mode]]["free_run"] = h
        print(f"[{key}] {MODES[mode]:5s} free run: {h}", flush=True)
os.makedirs(os.path.dirname(OUT), exist_ok=True)
with open(OUT, "w") as f:
    json.dump(report, f, indent=1)
print("written", OUT)
