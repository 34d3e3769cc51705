"""The array seam of the hot path with the reference's own signature.

``update_humans_parallel(type, agents_state, goals, obstacles, agents_params, dt, safety_space,
all_params_equal=False, last_is_robot=False) -> updated_state`` replaces the numba kernel of the same
name (reference: social_gym/src/forces_parallel.py:185-284); it accepts one world (``[N,13]``) exactly
like the reference, or a batch with a leading world axis (``[W,N,13]`` ...).  The work is done by
``cs_update_humans_parallel`` in libcrowdstep.so on the GPU -- there is no CPU path: without the HIP
library or a device the call raises.

Like the reference, the call mutates its inputs in place: ``goals`` rows are rotated when a goal is
reached, ``agents_state[:, 10:12]`` follows, and for the headed types ``agents_state[:, 3:5]`` receives
``R(theta) @ body_velocity`` (forces_parallel.py:231-234, 256).  Host arrays travel over PCIe on every
call; resident worlds (``batched.CrowdWorlds`` / ``MotionModelManager``) avoid that.
"""
from __future__ import annotations

import numpy as np

from ...batched import CrowdWorlds


def update_humans_parallel(type: int, agents_state: np.ndarray, goals: np.ndarray, obstacles, agents_params: np.ndarray,
                           dt: float, safety_space: np.ndarray, all_params_equal=False, last_is_robot=False):
    if type < 0 or type > 8:
        raise ValueError(f"Type {type} does not exist for this implementation")
    single = agents_state.ndim == 2
    S = agents_state[None] if single else agents_state
    G = goals[None] if single else goals
    P = agents_params
    saf = safety_space[None] if (single and safety_space is not None) else safety_space
    cw = CrowdWorlds(S, G, P, saf, obstacles, type=type, all_params_equal=all_params_equal, robot_row=last_is_robot)
    out_buf = cw.update_humans_parallel(dt, in_place=False)
    out = cw.get_states(out_buf).astype(agents_state.dtype)
    # in-place side effects on the caller's arrays
    s_in = cw.get_states()
    n = S.shape[1] - int(bool(last_is_robot))
    S[:, :n, 10:12] = s_in[:, :n, 10:12]
    if type >= 3:
        S[:, :n, 3:5] = s_in[:, :n, 3:5]
    mirror_goal_rotation(G, cw.get_goals())
    return out[0] if single else out


def mirror_goal_rotation(goals: np.ndarray, device_goals: np.ndarray) -> None:
    """Apply to the caller's (float64) goal lists the rotations the device performed on its float32
    copy: a row whose first goal changed is rotated left by one over its non-NaN prefix
    (forces_parallel.py:226-233); untouched rows keep their full-precision values."""
    g32 = goals.astype(np.float32)
    changed = np.any((g32[..., 0, :] != device_goals[..., 0, :]) & ~np.isnan(device_goals[..., 0, :]), axis=-1)
    for idx in zip(*np.nonzero(changed)):
        row = goals[idx]
        k = int(np.argmax(np.isnan(row).any(axis=1))) if np.isnan(row).any() else row.shape[0]
        # how far did it rotate? (one step per substep; fused blocks may rotate more than once)
        for shift in range(1, max(k, 1) + 1):
            cand = np.roll(row[:k], -shift, axis=0)
            if np.array_equal(cand.astype(np.float32), device_goals[idx][:k]):
                row[:k] = cand
                break
        else:  # e.g. respawn rewrote the goal: take the device values
            row[...] = device_goals[idx]
