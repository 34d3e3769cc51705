"""Per-substep accounting of the ORCA kernel's fast arithmetic (cs_worlds.orca_math = CS_ORCA_MATH_FAST / FMA: v_rcp / v_sqrt / v_rsq, mul + fma) against the
exact restatement (oracle/orca_oracle.c, float32 like RVO2; its own parity with rvo2 is UNPINNED -- the library is absent).

Protocol (the one `parity_util.fused_substeps_vs_oracle` uses for the force models): the trajectory is advanced by the EXACT
restatement; before every substep the GPU batch is re-synchronised to the restatement's rows and goal lists (same float32 bits),
steps ONE substep with the build under test, and its rows are compared with the restatement's rows of that substep.  Errors never
accumulate: every figure is the error one substep of the build adds from identical inputs.

RVO2's linear programmes are discontinuous functions of their inputs (a programme flips between feasible and infeasible, a
constraint enters or leaves the active set, the optimum is the intersection of two nearly parallel ORCA lines), so float32 does
not determine their result everywhere: the exact float32 restatement ITSELF is 1e-5 or more away from the same algorithm evaluated
in double, from the same rows, on ~1.8e-3 of the agent-substeps of a crossing (measured here, `beyond_bar_vs_f64_exact`).  An
agent-substep on which the build under test is beyond the bar (north_star: 1e-5 on positions / velocities) from the exact
restatement is therefore classified, in this order:
  f64    the exact restatement is itself >= bar away from the double evaluation (oracle/orca_oracle_f64.c) for this agent: float32
         does not resolve this agent-substep, the restatement's answer is one of several float32 answers;
  edge1  the exact restatement's own answer moves >= bar when the world's input rows move by ONE float32 ulp (random probes):
         a decision edge of the reference function at this input;
  edge4 / edge16  the same with up to 4 / 16 ulps (what a few roundings inside the programme amount to);
  op4    the exact restatement's own answer moves >= bar when each of its divisions, square roots and two-term product sums is perturbed
         by <= 4 float32 ulps (oracle/orca_oracle_probe.c, 128 random probes) -- what v_rcp / v_sqrt / v_rsq and mul + fma change;
  unexplained     none of these.
Every agent-substep the double evaluation does not explain is ALSO named by the probe (`decisions`): the first decision of RVO2's
computeNeighbors / computeNewVelocity / linearProgram1-3 that flips in a probe whose answer moved (e.g. "LP1: empty interval" = an LP2 -> LP3
feasibility flip, "LP2: line violated" = a constraint entering the active set), or "ill-conditioned intersection" when the answer moves with every
decision unchanged (the optimum is the intersection of two nearly parallel lines, or of a line and the speed circle it nearly touches);
`probe_reproduces_build` counts those on which some probe lands within the bar of the build's own answer.
Reported beside it: on the disagreeing agent-substeps, which of the two float32 answers is closer to the double one (a build that
were WORSE than float32 RVO2 would lose that vote), and the share of ALL agent-substeps beyond the bar from the double evaluation
for the build and for the exact restatement (the build is as far from real arithmetic as RVO2's own float32, no farther).
"""
from __future__ import annotations

import numpy as np

BAR = 1e-5
COLS = [0, 1, 3, 4]            # px, py, vx, vy  (north_star: positions / velocities)
PREF = [5, 6]                  # preferred velocity of the next substep (update_goals_orca)


def crossing(W, n, R, seed):
    from social_navigation_pyenvs_amd import scenarios as sc

    pos, yaw, g = sc.circular_crossing(W, n, R, seed)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    g = g.astype(np.float32)
    d = g[:, :, 0] - S[:, :, 0:2]
    S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    margin = np.full((W, n), 0.01, np.float32)
    return S, g, margin


def ulp_noise(rng, S, copies, ulps=1):
    """`copies` float32 copies of the rows S [rows, 13] with px, py, vx, vy each moved by an integer in [-ulps, ulps] float32 ulps"""
    out = np.repeat(S[None], copies, axis=0).copy()
    for c in COLS:
        x = out[:, :, c]
        step = rng.integers(-ulps, ulps + 1, size=x.shape)
        for _ in range(ulps):
            up = np.nextafter(x, np.float32(np.inf)); dn = np.nextafter(x, np.float32(-np.inf))
            x = np.where(step > 0, up, np.where(step < 0, dn, x))
            step = step - np.sign(step)
        out[:, :, c] = x
    return out


def edge_spread(orc, rng, S_w, g_w, margin_w, dt, probes, ulps=1):
    """per agent: how far the EXACT restatement's one-substep result (vx, vy) moves when the world's input rows are perturbed by `ulps` ulp"""
    base, _, _ = orc.orca_step_block(S_w[None], g_w[None], margin_w[None], dt, 1)
    noisy = ulp_noise(rng, S_w, probes, ulps)
    res, _, _ = orc.orca_step_block(noisy, np.repeat(g_w[None], probes, axis=0), np.repeat(margin_w[None], probes, axis=0), dt, 1)
    return np.abs(res[:, :, 3:5].astype(np.float64) - base[0][None, :, 3:5].astype(np.float64)).max(axis=(0, 2))


PROBE_ULPS, PROBE_COUNT = 4.0, 129
OD_LP2_CLIP = 7     # |prefVel|^2 > maxSpeed^2 with a unit preferred velocity and maxSpeed 1: a coin toss whose two branches agree to an ulp


def probe_classify(orc, S_w, margin_w, agent, dt, got_v, bar=BAR, seed=7):
    """(sensitive, label, reproduced): does the exact restatement's new velocity of `agent` move >= bar under <= PROBE_ULPS-ulp operation noise,
    the first decision that flips in such a probe (module docstring), and does some probe land within the bar of the build's answer got_v"""
    vel, tr = orc.orca_probe_agent(S_w, margin_w, agent, dt, k_ulps=PROBE_ULPS, probes=PROBE_COUNT, seed=seed)
    d = np.abs(vel[1:].astype(np.float64) - vel[0].astype(np.float64)).max(axis=1)
    moved = np.nonzero(d >= bar)[0]
    reproduced = bool((np.abs(vel[1:].astype(np.float64) - np.asarray(got_v, np.float64)).max(axis=1) < bar).any())
    if len(moved) == 0:
        return False, "not sensitive", reproduced
    votes = {}
    for p_ in moved + 1:
        k0, o0, _ = tr[0]; k1, o1, _ = tr[p_]
        L = min(len(k0), len(k1))
        label = "ill-conditioned intersection"
        for j in np.nonzero((k0[:L] != k1[:L]) | (o0[:L] != o1[:L]))[0]:
            if k0[j] != k1[j]:
                label = "trace diverged"; break
            if k0[j] == OD_LP2_CLIP:
                continue
            label = orc.ORCA_DECISION_KINDS[k0[j]]; break
        else:
            if len(k0) != len(k1):
                label = "trace diverged"
        votes[label] = votes.get(label, 0) + 1
    return True, max(votes, key=votes.get), reproduced


def substeps_vs_restatement(cw, S0, g0, margin, dt, n_substeps, *, bar=BAR, probes=48, seed=0, progress=None, max_examined=6000, collect=None):
    """Run the protocol above on the batch `cw` (a CrowdWorlds of type "orca" created from S0 / g0 / margin, any arithmetic mode).
    Returns a dict of figures.  Every agent-substep beyond the bar is classified, in this order:
    f64 / edge1 / edge4 / edge16 / unexplained (module docstring).  `collect` (a list): every beyond-bar agent-substep the double evaluation
    does not explain is appended as a record (substep, world, agent, the world's input rows / goals / margins, both results, its class)."""
    from oracle import crowd_oracle as orc

    rng = np.random.default_rng(seed)
    W, n = S0.shape[0], S0.shape[1]
    ref, rg = S0.copy(), g0.copy()
    dt32 = np.float32(dt)
    out = {"worlds": W, "agents": n, "substeps": n_substeps, "agent_substeps": 0, "beyond_bar": 0, "class_f64": 0, "class_edge1": 0, "class_edge4": 0, "class_edge16": 0,
           "class_op4": 0, "decisions": {}, "probe_reproduces_build": 0, "unexplained": 0, "worst": 0.0, "worst_within": 0.0, "worst_unexplained": 0.0, "bit_identical_agent_substeps": 0, "goal_column_flips": 0,
           "pref_velocity_worst": 0.0, "bar": bar, "probes": probes, "examined": 0, "not_examined": 0,
           "beyond_bar_vs_f64_build": 0, "beyond_bar_vs_f64_exact": 0, "disagree_build_closer_to_f64": 0, "disagree_exact_closer_to_f64": 0}
    hist = []
    for k in range(n_substeps):
        cw.set_states(ref); cw.set_goals(rg)
        cw.step(dt, 1)
        got = cw.get_states()
        nxt, ng, _ = orc.orca_step_block(ref, rg, margin, dt, 1)
        n64, _ = orc.orca_step_block_f64(ref, rg, margin, dt32, 1)
        g64 = got[..., COLS].astype(np.float64)
        err = np.abs(g64 - nxt[..., COLS].astype(np.float64)).max(axis=-1)          # [W, n]  build vs exact float32
        e64x = np.abs(nxt[..., COLS].astype(np.float64) - n64[..., COLS]).max(axis=-1)   # exact float32 vs double
        e64b = np.abs(g64 - n64[..., COLS]).max(axis=-1)                                 # build vs double
        assert np.all(np.isfinite(got[..., COLS])), f"substep {k + 1}: non-finite rows"
        out["agent_substeps"] += W * n
        out["bit_identical_agent_substeps"] += int(np.sum(np.all(got[..., COLS] == nxt[..., COLS], axis=-1)))
        out["beyond_bar_vs_f64_build"] += int(np.sum(e64b >= bar)); out["beyond_bar_vs_f64_exact"] += int(np.sum(e64x >= bar))
        hist.append(err.ravel())
        gflip = np.any(got[..., 10:12] != nxt[..., 10:12], axis=-1)
        out["goal_column_flips"] += int(gflip.sum())
        if gflip.any():      # a goal switch decided the other way: only on the strict < edge of update_goals (|goal - p| vs radius)
            ww, aa = np.nonzero(gflip)
            for w_, a_ in zip(ww, aa):
                p1 = nxt[w_, a_, 0:2].astype(np.float64)
                d = abs(np.linalg.norm(rg[w_, a_, 0].astype(np.float64) - p1) - float(ref[w_, a_, 8]))
                assert d < 1e-5, f"substep {k + 1} world {w_} agent {a_}: goal columns differ {d:.2e} away from the switch radius"
        same_goal = ~gflip
        pe = np.abs(got[..., PREF].astype(np.float64) - nxt[..., PREF].astype(np.float64)).max(axis=-1)
        out["pref_velocity_worst"] = max(out["pref_velocity_worst"], float(pe[same_goal & (err < bar)].max()))
        bad = err >= bar
        out["worst_within"] = max(out["worst_within"], float(err[~bad].max()) if (~bad).any() else 0.0)
        out["worst"] = max(out["worst"], float(err.max()))
        if bad.any():
            out["beyond_bar"] += int(bad.sum())
            out["disagree_build_closer_to_f64"] += int(np.sum(bad & (e64b < e64x))); out["disagree_exact_closer_to_f64"] += int(np.sum(bad & (e64x < e64b)))
            byf64 = bad & (e64x >= bar)
            out["class_f64"] += int(byf64.sum())
            rest = bad & ~byf64
            for w_ in np.nonzero(rest.any(axis=1))[0]:
                agents = np.nonzero(rest[w_])[0]
                if out["examined"] >= max_examined:
                    out["not_examined"] += len(agents)
                    continue
                out["examined"] += len(agents)
                spread = edge_spread(orc, rng, ref[w_], rg[w_], margin[w_], dt, probes, 1)
                left = [a_ for a_ in agents if spread[a_] < bar]
                out["class_edge1"] += len(agents) - len(left)
                cls = {int(a_): "edge1" for a_ in agents}
                probe = {int(a_): probe_classify(orc, ref[w_], margin[w_], int(a_), dt, got[w_, a_, 3:5], bar) for a_ in agents}
                for a_ in agents:
                    lab = probe[int(a_)][1]
                    out["decisions"][lab] = out["decisions"].get(lab, 0) + 1
                    out["probe_reproduces_build"] += int(probe[int(a_)][2])
                if left:
                    spread4 = edge_spread(orc, rng, ref[w_], rg[w_], margin[w_], dt, probes, 4)
                    left16 = [a_ for a_ in left if spread4[a_] < bar]
                    out["class_edge4"] += len(left) - len(left16)
                    for a_ in left:
                        cls[int(a_)] = "edge4"
                    if left16:
                        spread16 = edge_spread(orc, rng, ref[w_], rg[w_], margin[w_], dt, probes, 16)
                        for a_ in left16:
                            if spread16[a_] >= bar:
                                out["class_edge16"] += 1
                                cls[int(a_)] = "edge16"
                            elif probe[int(a_)][0]:
                                out["class_op4"] += 1
                                cls[int(a_)] = "op4"
                            else:
                                out["unexplained"] += 1
                                cls[int(a_)] = "unexplained"
                                out["worst_unexplained"] = max(out["worst_unexplained"], float(err[w_, a_]))
                if collect is not None:
                    for a_ in agents:
                        collect.append(dict(substep=k, world=int(w_), agent=int(a_), cls=cls[int(a_)], decision=probe[int(a_)][1], err=float(err[w_, a_]), S=ref[w_].copy(), g=rg[w_].copy(),
                                            margin=margin[w_].copy(), got=got[w_, a_].copy(), exact=nxt[w_, a_].copy(), f64=n64[w_, a_].copy()))
        ref, rg = nxt, ng
        if progress and (k + 1) % progress == 0:
            print(f"  substep {k + 1}/{n_substeps}: beyond bar {out['beyond_bar']} (f64 {out['class_f64']}, edge1 {out['class_edge1']}, edge4 {out['class_edge4']}, edge16 {out['class_edge16']}, op4 {out['class_op4']}, "
                  f"unexplained {out['unexplained']}), worst {out['worst']:.2e}", flush=True)
    allerr = np.concatenate(hist)
    out["p50"], out["p99"], out["p9999"] = (float(np.quantile(allerr, q)) for q in (0.5, 0.99, 0.9999))
    for k_ in ("beyond_bar", "beyond_bar_vs_f64_build", "beyond_bar_vs_f64_exact", "unexplained"):
        out[k_ + "_share"] = out[k_] / max(1, out["agent_substeps"])
    out["mean_displacement_m"] = float(np.mean(np.linalg.norm(ref[..., 0:2] - S0[..., 0:2], axis=-1)))
    return out


def free_run_health(cw_factory, S0, g0, margin, dt, n_substeps, fused=20):
    """Free-running crowd (no re-synchronisation): what the episode looks like under a build -- overlaps, speeds, progress.  Two builds'
    figures are compared by the caller (a chaotic system: rows diverge, statistics must not)."""
    cw = cw_factory()
    n = S0.shape[1]
    worst_overlap, over = 0.0, 0
    eye = 10.0 * np.eye(n)[None]
    rad = S0[:, :, 8] + margin
    rsum = rad[:, :, None] + rad[:, None, :]
    for _ in range(n_substeps // fused):
        cw.step(dt, fused)
        S = cw.get_states()
        dd = np.linalg.norm(S[:, :, None, 0:2] - S[:, None, :, 0:2], axis=-1) + eye
        pen = np.maximum(0.0, rsum - dd)
        worst_overlap = max(worst_overlap, float(pen.max()))
        over += int(np.sum(pen > 1e-3) // 2)
    S = cw.get_states()
    return {"worst_overlap_m": worst_overlap, "overlapping_pairs_sampled": over, "max_speed_over_vmax": float((np.linalg.norm(S[..., 3:5], axis=-1) - S[..., 12]).max()),
            "mean_displacement_m": float(np.mean(np.linalg.norm(S[..., 0:2] - S0[..., 0:2], axis=-1))),
            "mean_goal_distance_m": float(np.mean(np.linalg.norm(S[..., 10:12] - S[..., 0:2], axis=-1)))}
