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
