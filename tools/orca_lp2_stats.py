#!/usr/bin/env python3
"""Diagnostic (CPU, restatement): how many agents of the dense phase of the cfg4 crossing violate a line in linearProgram2, how many
linearProgram1 calls each makes and how long their inner loops are -- the numbers behind HISTORY.md Part II 4.2 (lp2 on lane groups: rejected)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import crowd_oracle as orc
from social_navigation_pyenvs_amd import scenarios as sc
W, n = 16, 25
pos, yaw, g = sc.circular_crossing(W, n, 7.0, 1000)
S = sc.make_states(pos, yaw, g).astype(np.float32)
d = g[:, :, 0] - S[:, :, 0:2]
S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
margin = np.full((W, n), 0.01, np.float32)
ref, rg = S, g.astype(np.float32)
for step in range(30):
    ref, rg, _ = orc.orca_step_block(ref, rg, margin, 0.0125, 20)
tot_agents = viol_agents = lp1_calls = inner = 0
maxcalls = []
for w in range(W):
    v, lines, nl = orc.orca_new_velocities(ref[w, :, 0:2], ref[w, :, 3:5], ref[w, :, 5:7], ref[w, :, 8] + 0.01, ref[w, :, 12], time_step=0.0125, return_lines=True)
    for a in range(n):
        L = lines[a, :nl[a]].astype(np.float32)
        pv = ref[w, a, 5:7].astype(np.float32); vmax = np.float32(ref[w, a, 12])
        r = pv.copy()
        if r @ r > vmax * vmax: r = r / np.sqrt(r @ r) * vmax
        calls = 0
        for i in range(len(L)):
            p, dr = L[i, :2], L[i, 2:]
            if dr[0] * (p[1] - r[1]) - dr[1] * (p[0] - r[0]) > 0:
                calls += 1; inner += i
                dot = p @ dr; disc = dot * dot + vmax * vmax - p @ p
                if disc < 0: break
                sq = np.sqrt(disc); tL, tR = -dot - sq, -dot + sq; fail = False
                for j in range(i):
                    pj, dj = L[j, :2], L[j, 2:]
                    den = dr[0] * dj[1] - dr[1] * dj[0]; num = dj[0] * (p[1] - pj[1]) - dj[1] * (p[0] - pj[0])
                    if abs(den) <= 1e-5:
                        if num < 0: fail = True; break
                        continue
                    t = num / den
                    if den >= 0: tR = min(tR, t)
                    else: tL = max(tL, t)
                    if tL > tR: fail = True; break
                if fail: break
                t = dr @ (pv - p); t = min(max(t, tL), tR); r = p + t * dr
        tot_agents += 1; viol_agents += calls > 0; lp1_calls += calls
        maxcalls.append(calls)
mc = np.array(maxcalls).reshape(W, n)
print("agents", tot_agents, "with >=1 LP1 in LP2:", viol_agents / tot_agents, "LP1 calls per such agent:", lp1_calls / max(viol_agents, 1), "inner iterations per call:", inner / max(lp1_calls, 1))
print("per pair of worlds (one wavefront): max calls", np.mean([mc[i:i+2].max() for i in range(0, W, 2)]), "sum calls", np.mean([mc[i:i+2].sum() for i in range(0, W, 2)]))
