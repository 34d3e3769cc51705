#!/usr/bin/env python3
"""Diagnostic (CPU, restatement): linearProgram3 of the dense cfg4 crossing replayed per wavefront (two worlds) under two schedules of the
lane groups -- the kernel's (one level per agent and round; a pass runs until its slowest group is through) and re-dealing the agents after
every linearProgram1 call -- with instruction estimates per round / pass / iteration: HISTORY.md Part II 4.2 (the second schedule is no gain)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import crowd_oracle as orc
from social_navigation_pyenvs_amd import scenarios as sc
EPS = 1e-5
def det(a, b): return a[0]*b[1] - a[1]*b[0]
def lp1(L, k, radius, opt, dirOpt):
    p, d = L[k][:2], L[k][2:]
    dot = p @ d; disc = dot*dot + radius*radius - p @ p
    if disc < 0: return None
    sq = np.sqrt(disc); tL, tR = -dot - sq, -dot + sq
    for j in range(k):
        pj, dj = L[j][:2], L[j][2:]
        den = det(d, dj); num = det(dj, p - pj)
        if abs(den) <= EPS:
            if num < 0: return None
            continue
        t = num/den
        if den >= 0: tR = min(tR, t)
        else: tL = max(tL, t)
        if tL > tR: return None
    if dirOpt: t = tR if opt @ d > 0 else tL
    else: t = min(max(d @ (opt - p), tL), tR)
    return p + t*d
def lp2(L, radius, opt, dirOpt):
    calls = 0
    if dirOpt: r = opt*radius
    elif opt @ opt > radius*radius: r = opt/np.sqrt(opt @ opt)*radius
    else: r = opt.copy()
    for i in range(len(L)):
        if det(L[i][2:], L[i][:2] - r) > 0:
            calls += 1
            nr = lp1(L, i, radius, opt, dirOpt)
            if nr is None: return i, r, calls
            r = nr
    return len(L), r, calls
def lp3_levels(L, begin, radius, r):
    dist = 0.0; levels = []
    for i in range(begin, len(L)):
        if det(L[i][2:], L[i][:2] - r) > dist:
            P = []
            for j in range(i):
                d_ = det(L[i][2:], L[j][2:])
                if abs(d_) <= EPS:
                    if L[i][2:] @ L[j][2:] > 0: continue
                    pt = 0.5*(L[i][:2] + L[j][:2])
                else:
                    pt = L[i][:2] + det(L[j][2:], L[i][:2] - L[j][:2])/d_ * L[i][2:]
                dr = L[j][2:] - L[i][2:]; dr = dr/np.sqrt(dr @ dr)
                P.append(np.concatenate([pt, dr]))
            P = np.array(P).reshape(-1, 4)
            f, nr, calls = lp2(P, radius, np.array([-L[i][3], L[i][2]]), True)
            if f >= len(P): r = nr
            levels.append((i, calls))
            dist = det(L[i][2:], L[i][:2] - r)
    return levels
W, n = 16, 25
pos, yaw, g = sc.circular_crossing(W, n, 7.0, 1000)
S = sc.make_states(pos, yaw, g).astype(np.float32)
d = g[:, :, 0] - S[:, :, 0:2]; S[:, :, 5:7] = d/np.linalg.norm(d, axis=-1, keepdims=True)
margin = np.full((W, n), 0.01, np.float32)
ref, rg = S, g.astype(np.float32)
for step in range(30): ref, rg, _ = orc.orca_step_block(ref, rg, margin, 0.0125, 20)
cur_slots = new_slots = 0; waves = 0; stats = []
for w0 in range(0, W, 2):
    agents = []
    for w in (w0, w0 + 1):
        v, lines, nl = orc.orca_new_velocities(ref[w, :, 0:2], ref[w, :, 3:5], ref[w, :, 5:7], ref[w, :, 8] + 0.01, ref[w, :, 12], time_step=0.0125, return_lines=True)
        for a in range(n):
            L = lines[a, :nl[a]].astype(np.float64); vmax = float(ref[w, a, 12])
            f, r, _ = lp2(L, vmax, ref[w, a, 5:7].astype(np.float64), False)
            if f < len(L): agents.append(lp3_levels(L, f, vmax, r))
    stats += agents
    # current schedule: rounds by level; per round: pending agents' next level; passes of 8; pass iterations = max(calls)+1
    it_cur = 0; passes_cur = 0; rounds_cur = 0
    rnd = 0
    while True:
        pend = [lv[rnd] for lv in agents if len(lv) > rnd]
        if not pend: break
        rounds_cur += 1
        pa = [c for (i, c) in pend if i <= 8]; pb = [c for (i, c) in pend if i == 9]
        for lst, gsz in ((pa, 8), (pb, 4)):
            for k in range(0, len(lst), gsz):
                passes_cur += 1; it_cur += max(lst[k:k+gsz]) + 1
        rnd += 1
    # new schedule: rounds by LP1 iteration
    seq = []   # per agent: list of actions: 'P' (projection) then calls+1 iterations per level
    for lv in agents:
        s = []
        for (i, c) in lv: s += ['P'] + ['I']*(c + 1)
        seq.append(s)
    t = 0; passes_P = passes_I = rounds_new = 0
    pos_ = [0]*len(seq)
    while any(p < len(s) for p, s in zip(pos_, seq)):
        rounds_new += 1
        # projection for those at 'P'
        nP = sum(1 for p, s in zip(pos_, seq) if p < len(s) and s[p] == 'P')
        if nP: passes_P += -(-nP // 8)
        pos_ = [p + 1 if p < len(s) and s[p] == 'P' else p for p, s in zip(pos_, seq)]
        nI = sum(1 for p, s in zip(pos_, seq) if p < len(s) and s[p] == 'I')
        if nI: passes_I += -(-nI // 8)
        pos_ = [p + 1 if p < len(s) and s[p] == 'I' else p for p, s in zip(pos_, seq)]
    print(f"wave {w0//2}: infeasible {len(agents)}; CURRENT rounds {rounds_cur} passes {passes_cur} iteration-slots {it_cur} -> ~{rounds_cur*90 + passes_cur*100 + it_cur*80} instr | NEW rounds {rounds_new} projection passes {passes_P} LP1 passes {passes_I} -> ~{rounds_new*60 + passes_P*75 + passes_I*105} instr")
lv_per = np.mean([len(a) for a in stats]); calls = np.mean([sum(c for _, c in a) for a in stats])
print("agents", len(stats), "levels/agent", lv_per, "LP1 calls/agent", calls, "max calls in a level", max(c for a in stats for _, c in a))
