"""Shared helpers of the parity tests: per-column comparison of state rows (no fixture is skipped), and a per-group
record of how many cases meet north_star's 1e-5 that the GPU session writes to gpurun_out/parity_report.json.

Column rules for one Euler substep (or a block of substeps) of a row  px,py,theta,vx,vy,bvx,bvy,omega :
  * px, py, bvx, bvy : absolute, always.  They do not depend on the new heading or the new omega.
  * theta, vx, vy    : absolute, with the tolerance widened by what float32 can hold of theta + omega_in * dt:
                       tol + 4 * eps32 * |omega_in * dt|.  (vx, vy) = R(theta_new) * bv.  A row whose incoming
                       |omega * dt| exceeds 1e4 rad has no float32 heading at all (the reference's own explicit Euler
                       on omega has diverged there: |omega| 1e11 .. 1e37 in four golden episode cases); its three
                       heading columns are reported as `unrepresentable`, everything else in that row is still compared.
  * omega            : RELATIVE (|omega_out| reaches 4e6 in the golden vectors when forces are large), skipped only
                       where float32 overflows (|omega| > 1e30).
"""
from __future__ import annotations

import json
import os

import numpy as np

EPS32 = float(np.finfo(np.float32).eps)
PV = [0, 1, 3, 4]
REPORT: dict[str, dict] = {}


def f32(a):
    return None if a is None else np.asarray(a, dtype=np.float32)


def compare_rows(got, ref, omega_in, dt, tol, headed, what="", omega_rtol=2e-4):
    """Assert the column rules above for rows got / ref [n, >=8].  Returns (worst absolute error over the compared
    position / velocity entries, number of rows whose heading columns are unrepresentable in float32)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    n = ref.shape[0]
    omega_in = np.abs(np.asarray(omega_in, dtype=np.float64)[:n])
    e_pos = np.abs(got[:, [0, 1]] - ref[:, [0, 1]])
    assert np.all(np.isfinite(got[:, [0, 1]])) and e_pos.max() < tol, f"{what}: position error {e_pos.max():.3e} (tol {tol:.1e})"
    worst = float(e_pos.max())
    unrepresentable = 0
    if not headed:
        e_v = np.abs(got[:, [3, 4]] - ref[:, [3, 4]])
        assert e_v.max() < tol, f"{what}: velocity error {e_v.max():.3e} (tol {tol:.1e})"
        return max(worst, float(e_v.max())), 0
    e_b = np.abs(got[:, [5, 6]] - ref[:, [5, 6]])
    assert np.all(np.isfinite(got[:, [5, 6]])) and e_b.max() < tol, f"{what}: body velocity error {e_b.max():.3e} (tol {tol:.1e})"
    worst = max(worst, float(e_b.max()))
    swing = omega_in * abs(dt)
    ok = swing <= 1e4
    unrepresentable = int(np.sum(~ok))
    row_tol = tol + 4.0 * EPS32 * swing
    if np.any(ok):
        dth = np.abs(got[ok, 2] - ref[ok, 2])
        dth = np.minimum(dth, np.abs(dth - 2.0 * np.pi))          # +-pi is one heading (wrap_angle may sit on either side)
        assert np.all(dth < np.maximum(row_tol[ok], 2e-6)), f"{what}: heading error {dth.max():.3e}"
        e_v = np.abs(got[ok][:, [3, 4]] - ref[ok][:, [3, 4]])
        assert np.all(e_v < row_tol[ok, None]), f"{what}: velocity error {e_v.max():.3e} (tol {row_tol[ok].min():.1e}..{row_tol[ok].max():.1e})"
        calm = ok & (swing < 1.0)
        if np.any(calm):
            worst = max(worst, float(np.abs(got[calm][:, [3, 4]] - ref[calm][:, [3, 4]]).max()))
    small = (np.abs(ref[:, 7]) < 1e30) & (omega_in < 1e30)
    if np.any(small):
        rel = np.abs(got[small, 7] - ref[small, 7]) / np.maximum(1.0, np.abs(ref[small, 7]))
        assert np.all(np.isfinite(got[small, 7])) and rel.max() < omega_rtol, f"{what}: omega relative error {rel.max():.3e}"
    return worst, unrepresentable


def record(group: str, err: float, bar: float = 1e-5, unrepresentable_rows: int = 0) -> None:
    """One compared case of `group` with worst absolute position / velocity error `err`."""
    r = REPORT.setdefault(group, {"cases": 0, "within_1e-5": 0, "worst": 0.0, "unrepresentable_heading_rows": 0})
    r["cases"] += 1
    r["within_1e-5"] += int(err < bar)
    r["worst"] = max(r["worst"], float(err))
    r["unrepresentable_heading_rows"] += int(unrepresentable_rows)


def write_report(root: str) -> None:
    if not REPORT:
        return
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


# ---------------------------------------------------------------------------------------------------------------------
# Per-substep parity INSIDE a fused launch (cs_step_trace): substep k + 1 of the GPU against the oracle's single substep
# restarted from the GPU's own substep-k rows.  The bar is north_star's 1e-5 on every substep; a substep on which the
# float32 instantiation of the oracle itself is farther than that from its float64 instantiation (a body contact at
# 25 kN/m after a respawn, Moussaid's sign(theta ~ 0)) is ill-conditioned in float32 whoever computes it: there the GPU
# must stay within F32_SLACK times the float32 oracle's own error, and the case is counted as outside 1e-5 in the report.
# F32_SLACK = 3: the float32 oracle's error is ONE realisation of float32 rounding (libm exp / sqrt, IEEE divide); the kernel's
# is another, with 1-ulp v_exp / v_rsq / v_rcp and log2|A| folded into the exponent argument (|x| up to 11 larger: 1.4 x the
# argument rounding).  Worst measured ratio: 2.6 (g1_direct case 177, hsfm_new with 1e7 N overlaps), recorded in the report.
# ---------------------------------------------------------------------------------------------------------------------
F32_SLACK = 3.0


def single_call_bar(ref32, ref64, omega_in, dt, headed, bar=1e-5):
    """The tolerance of ONE library call against the f64 oracle from the same float32 inputs: north_star's 1e-5, or -- where the float32
    instantiation of the oracle is itself farther than that from the float64 one -- F32_SLACK times the float32 oracle's own error,
    measured for this very case (no blanket tolerance).  Returns (tolerance, float32 oracle error)."""
    e32 = float(row_errors(ref32, ref64, omega_in, dt, headed)[0].max())
    return max(bar, F32_SLACK * e32), e32


def row_errors(got, ref, omega_in, dt, headed, theta_in=None):
    """Vectorised column rules of this module for rows [..., >= 8]: returns (err [...], omega_rel [...], lost [...] bool):
    err = worst of |d px|, |d py| (+ |d bvx|, |d bvy|, and theta / vx / vy minus what float32 holds of theta + omega_in dt when
    headed); omega_rel = relative omega error; lost = rows whose heading float32 cannot hold at all.
    `theta_in` (headed models): the INCOMING headings.  The reference's bound_angle leaves every heading in [-pi, pi]; a float32 row whose
    incoming heading lies outside lost it in an EARLIER substep (omega dt of 1e17 rad, hsfm_new* driven into contact: the reference's own
    omega reaches 1e108 there) -- its rotation R(theta) is then only good to ulp(theta) ~ 5e-5 rad, and with it the body-frame
    velocity: such a row is compared on its position only and counted as lost."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        d = np.abs(got[..., :8] - ref[..., :8])
        e = np.maximum(d[..., 0], d[..., 1])
        zero = np.zeros(e.shape)
        if not headed:
            return np.maximum(e, np.maximum(d[..., 3], d[..., 4])), zero, np.zeros(e.shape, bool)
        e = np.maximum(e, np.maximum(d[..., 5], d[..., 6]))
        swing = np.abs(np.asarray(omega_in, dtype=np.float64)) * abs(dt)
        lost = ~(swing <= 1e4)
        gone = np.zeros(e.shape, bool) if theta_in is None else ~(np.abs(np.asarray(theta_in, dtype=np.float64)) <= np.pi * (1 + 1e-6))
        e = np.where(gone, np.maximum(d[..., 0], d[..., 1]), e)
        lost = lost | gone
        allow = 4.0 * EPS32 * np.where(lost, 0.0, swing)
        dth = np.minimum(d[..., 2], np.abs(d[..., 2] - 2.0 * np.pi))
        hv = np.maximum(np.maximum(d[..., 3], d[..., 4]), dth) - allow
        e = np.where(lost, e, np.maximum(e, hv))
        small = (np.abs(ref[..., 7]) < 1e30) & (np.abs(np.asarray(omega_in, dtype=np.float64)) < 1e30)
        rel = np.where(small, d[..., 7] / np.maximum(1.0, np.abs(ref[..., 7])), 0.0)
    e = np.where(np.isfinite(e), e, np.inf)
    return e, rel, lost


def moussaid_sign_ambiguous(Sw, P0, n, tol=2e-6):
    """Rows of ONE world Sw [rows, 13] with a Moussaid partner whose theta_ij (forces_parallel.py:120-124) lies within `tol` rad of 0:
    the force law multiplies a full-size lateral term by sign(theta_ij), and no float32 evaluation (the reference's own float64
    follows its rounding noise there too) decides that sign reliably -- such a row is a coin flip, not a parity
    case.  Returns a bool mask [n]."""
    p, v = Sw[:, 0:2], Sw[:, 3:5]
    lam = float(P0[12])
    amb = np.zeros(n, bool)
    for i in range(n):
        d = p[i] - np.delete(p, i, axis=0)
        dist = np.linalg.norm(d, axis=1)
        nn = d / dist[:, None]
        w = lam * (v[i] - np.delete(v, i, axis=0)) - nn
        ii = w / np.linalg.norm(w, axis=1)[:, None]
        cross = ii[:, 1] * nn[:, 0] - ii[:, 0] * nn[:, 1]
        dot = -(ii[:, 0] * nn[:, 0] + ii[:, 1] * nn[:, 1])
        th = np.arctan2(cross, dot)
        # only partners whose force matters at the bar: Ei e^{-dist / F} > 1e-3 N-ish is not needed -- any partner on the edge counts
        amb[i] = bool(np.any(np.abs(th) < tol))
    return amb


def _unicycle(rb, A, dt):
    """robot.step(ActionRot(v, r), dt) (robot_agent.py:119-136) on [w, 13] rows in float64"""
    rb = np.array(rb, dtype=np.float64, copy=True)
    v, r = A[:, 0], A[:, 1]
    rb[:, 0] += np.cos(rb[:, 2] + r) * v * dt
    rb[:, 1] += np.sin(rb[:, 2] + r) * v * dt
    rb[:, 2] = np.mod(rb[:, 2] + r, 2 * np.pi)
    rb[:, 3] = np.cos(rb[:, 2]) * v
    rb[:, 4] = np.sin(rb[:, 2]) * v
    return rb


def fused_substeps_vs_oracle(cw, type_, S0, goals0, P, safety, obstacles, dt, nsub, peq, *, respawn=None, respawn_bounds=None,
                             robot_row=False, robot=None, action=None, worlds=None, group=None, what="", bar=1e-5, omega_rtol=2e-4, kinematics=0):
    """Run cw.step_trace(dt, nsub) (cs_step's kernel build, state updated in place) and check EVERY substep of the fused
    launch: record k + 1 against orc.step_block(1 substep, respawn rule included) from the GPU's record k (float32 rows
    read as float64).  S0 [W, rows, 13], goals0 [W, n, G, 2] = what the batch was created from; `worlds` = the worlds to
    check (default all).  `robot` [W, 13] + `action` [W, 2] (with robot_row): the visible robot of a Gym step, moved by its action
    before every substep (social_nav_gym.py:240-243) and handed to the crowd as the last state row.  `kinematics` = 1: the action rows are
    ActionRot (v, r) (the batch has cw.unicycle set; robot_agent.py:119-136) and the ROBOT's own records are checked too: record k against one
    float64 unicycle step from record k - 1 (positions / velocities 2e-6, the yaw 1e-6: one float32 step with the hardware sin / cos).
    Returns a dict of figures; asserts the bar described above on every substep of every checked world."""
    from oracle import crowd_oracle as orc

    trace = cw.step_trace(dt, nsub, action) if action is not None else cw.step_trace(dt, nsub)     # [K, W, rows, 12]
    S0 = np.asarray(S0, dtype=np.float32); goals0 = np.asarray(goals0, dtype=np.float32)
    if S0.ndim == 2:
        S0, goals0 = S0[None], goals0[None]
    W, rows = S0.shape[0], S0.shape[1]
    n = rows - int(robot_row)
    sel = np.arange(W) if worlds is None else np.asarray(worlds)
    S = S0[sel].astype(np.float64)
    goals = goals0[sel].astype(np.float64).reshape(len(sel), n, -1, 2)
    G = goals.shape[2]
    P64 = np.asarray(P, dtype=np.float32).astype(np.float64)
    Psel = P64 if P64.ndim == 2 else P64[sel]
    saf = np.ascontiguousarray(np.broadcast_to(np.asarray(0.0 if safety is None else safety, dtype=np.float32), (W, rows)))[sel].astype(np.float64)
    obs = None if obstacles is None else np.asarray(obstacles, dtype=np.float32).astype(np.float64)
    rsp = np.zeros(len(sel), bool) if respawn is None else np.broadcast_to(np.asarray(respawn).astype(bool), (W,))[sel]
    R = None if robot is None else np.asarray(robot, dtype=np.float32).reshape(W, 13)[sel].astype(np.float64)
    if R is not None:   # the robot row as substep 1 sees it: robot.step(action, dt) in float32 (robot_agent.py:114-118, holonomic)
        S[:, n] = R
        if action is not None:
            A = np.ascontiguousarray(np.broadcast_to(np.asarray(action, dtype=np.float32), (W, 2)))[sel].astype(np.float64)
            if kinematics == 0:
                S[:, n, 0:2] = (R[:, 0:2] + A * float(np.float32(dt))).astype(np.float32)
                S[:, n, 3:5] = A
            else:
                S[:, n] = _unicycle(R, A, float(np.float32(dt))).astype(np.float32)
    headed = type_ >= 3
    out = {"substeps": 0, "within": 0, "worst": 0.0, "worst_f32_oracle": 0.0, "ill_conditioned": 0, "goal_flips": 0, "lost_heading_rows": 0}
    for k in range(nsub):
        ref = np.empty_like(S); ref32 = np.empty_like(S); gnext = np.empty_like(goals)
        for flag in (False, True):
            m = rsp == flag
            if not m.any():
                continue
            rp = (respawn_bounds[0], respawn_bounds[1], 0.0) if flag else (0.0, 0.0, 0.0)
            o = obs if (obs is None or obs.ndim == 4) else obs[sel][m]
            args = (type_, S[m], goals[m], o, Psel if Psel.ndim == 2 else Psel[m], dt, 1, saf[m], peq)
            # (the robot row of S is already the robot as THIS substep sees it -- the GPU's own float32 robot motion -- so the oracle
            #  takes it as a standing row: a robot 1e-6 m off would be 2e-5 m/s on a human it touches, k1 = 120 kN/m)
            kw = dict(robot_visible=robot_row, respawn=flag, respawn_par=rp, robot=None if R is None else S[m][:, n], action=None)
            r64, g64, _ = orc.step_block(*args, **kw)
            with np.errstate(over="ignore", invalid="ignore"):
                r32, _, _ = orc.step_block(*args, dtype=np.float32, **kw)
            ref[m], ref32[m], gnext[m] = r64, r32, g64.reshape(-1, n, G, 2)
        got = trace[k][sel][:, :n].astype(np.float64)                    # [w, n, 12]
        om_in = S[:, :n, 7]
        e_gpu, rel, lost = row_errors(got, ref[:, :n], om_in, dt, headed, theta_in=S[:, :n, 2])
        e_f32, _, _ = row_errors(ref32[:, :n], ref[:, :n], om_in, dt, headed, theta_in=S[:, :n, 2])
        if type_ % 3 == 2 and np.any(e_gpu >= bar):
            # Moussaid: a row that fails may sit on sign(theta_ij ~ 0) (also as the partner of such a pair: the reaction -f flips with it)
            for a_ in np.nonzero((e_gpu >= bar).any(axis=1))[0]:
                amb = moussaid_sign_ambiguous(S[a_], (Psel if Psel.ndim == 2 else Psel[a_])[0], n)
                if amb.any():
                    out["sign_ambiguous_rows"] = out.get("sign_ambiguous_rows", 0) + int((e_gpu[a_] >= bar).sum())
                    e_gpu[a_] = np.where(e_gpu[a_] >= bar, 0.0, e_gpu[a_])
        wg, wf = e_gpu.max(axis=1), e_f32.max(axis=1)                     # per world
        allowed = np.maximum(bar, F32_SLACK * wf)
        bad = ~(wg < allowed)
        assert not bad.any(), (f"{what}: substep {k + 1} of the fused launch, world {int(sel[np.argmax(bad)])}: GPU {wg[bad].max():.3e} vs f64 oracle "
                               f"(float32 oracle {wf[bad].max():.3e}, bar {bar:.0e})")
        if headed:
            relw = np.where(lost, 0.0, rel).max(axis=1)
            assert np.all(relw < np.maximum(omega_rtol, 1e3 * wf)), f"{what}: substep {k + 1}: omega relative error {relw.max():.3e}"
        # exact / control-flow columns: the state's goal columns, the head of the goal list
        gcol = np.abs(got[..., 8:10] - ref[:, :n, 10:12]).max(axis=-1)
        head = np.abs(np.nan_to_num(got[..., 10:12]) - np.nan_to_num(gnext[:, :, 0])).max(axis=-1)
        flip = (head > 1e-5) | (gcol > 1e-5)
        if flip.any():                                                    # a goal-switch / respawn decision on a float32 rounding edge
            fw, fi = np.nonzero(flip)
            for a_, b_ in zip(fw, fi):
                p_in, r_in = S[a_, b_, 0:2], S[a_, b_, 8]
                edge = abs(np.linalg.norm(goals[a_, b_, 0] - p_in) - r_in) < 1e-5 or abs(np.linalg.norm(p_in - goals[a_, b_, 0]) - 3.0) < 1e-4
                assert edge, f"{what}: substep {k + 1} world {int(sel[a_])} human {b_}: goal columns differ away from a decision edge"
                if G >= 2 and np.allclose(got[a_, b_, 10:12], gnext[a_, b_, 1], atol=1e-5):
                    gnext[a_, b_, [0, 1]] = gnext[a_, b_, [1, 0]]
            out["goal_flips"] += int(flip.sum())
        out["substeps"] += len(sel); out["within"] += int(np.sum(wg < bar)); out["ill_conditioned"] += int(np.sum(wf >= bar))
        out["worst"] = max(out["worst"], float(wg.max())); out["worst_f32_oracle"] = max(out["worst_f32_oracle"], float(wf.max()))
        out["lost_heading_rows"] += int(lost.sum())
        if group:
            for x, y in zip(wg, wf):
                record(group, float(x), bar)
                REPORT[group]["worst_f32_oracle"] = max(REPORT[group].get("worst_f32_oracle", 0.0), float(y))
        # the next substep starts from the GPU's rows; goal lists follow the oracle's rotation with the GPU's (float32) head
        S[:, :n, 0:8] = got[..., 0:8]
        S[:, :n, 10:12] = got[..., 8:10]
        if robot_row and trace.shape[2] > n:
            if kinematics == 1 and R is not None and action is not None:
                # the robot's record k = the robot as substep k + 2 sees it (after its next move), the last one as it stands at the end
                want = _unicycle(S[:, n], A, float(np.float32(dt))) if k + 1 < nsub else S[:, n]
                rgot = trace[k][sel][:, n].astype(np.float64)
                assert np.abs(rgot[:, [0, 1, 3, 4]] - want[:, [0, 1, 3, 4]]).max() < 2e-6, f"{what}: substep {k + 1}: the unicycle robot's record"
                dyaw = np.abs(rgot[:, 2] - want[:, 2]); dyaw = np.minimum(dyaw, 2 * np.pi - dyaw)
                assert dyaw.max() < 1e-6 and np.all((rgot[:, 2] >= 0) & (rgot[:, 2] < 2 * np.pi + 1e-6)), f"{what}: substep {k + 1}: the unicycle robot's yaw"
                out["robot_records"] = out.get("robot_records", 0) + len(sel)
            S[:, n, 0:8] = trace[k][sel][:, n, 0:8]                       # the robot as the next substep sees it (GPU record)
        goals = gnext
        keep = ~np.isnan(goals[:, :, 0, 0])
        goals[:, :, 0][keep] = got[..., 10:12][keep]
    # coin flips of Moussaid's sign(theta ~ 0) are rare events, not a way out: at most 0.2 % of the rows checked
    assert out.get("sign_ambiguous_rows", 0) <= max(2, 0.002 * out["substeps"] * n), (what, out)
    return out
