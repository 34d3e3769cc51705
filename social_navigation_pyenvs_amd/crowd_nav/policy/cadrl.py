"""Device versions of the array helpers CADRL / SARL / LSTM-RL call once per robot decision
(reference: crowd_nav/policy/cadrl.py:13-105).  Only the pure-array functions live here -- the policies
themselves (value networks, training) are consumers of the env and stay the user's code."""
from __future__ import annotations

import ctypes as C
import itertools

import numpy as np

from ... import _lib
from ..._lib import DeviceBuffer, check


def build_action_space_array(v_pref: float, speed_samples: int = 5, rotation_samples: int = 16) -> np.ndarray:
    """The holonomic action set of CADRL.build_action_space (cadrl.py:181-206): (0, 0) + rotations x speeds."""
    speeds = [(np.exp((i + 1) / speed_samples) - 1) / (np.e - 1) * v_pref for i in range(speed_samples)]
    rotations = np.linspace(0, 2 * np.pi, rotation_samples, endpoint=False)
    acts = [[0.0, 0.0]] + [[s * np.cos(r), s * np.sin(r)] for r, s in itertools.product(rotations, speeds)]
    return np.array(acts, dtype=np.float64)


def compute_rotated_states_and_reward(action_space, next_humans_state, current_humans_state, current_robot_state, dt,
                                      theta_and_omega_visible=False):
    """Same signature and return as the reference function (cadrl.py:42-83): (rotated_states [A, N, 13|15],
    rewards [A]).  A leading world axis on the three state arrays batches W robots:
    next [W,N,4|6], current [W,N,5|7], robot [W,8+] -> ([W,A,N,13|15], [W,A])."""
    _lib.require_gpu()
    cur = np.asarray(current_humans_state)
    single = cur.ndim == 2
    f = np.float32
    acts = np.ascontiguousarray(action_space, dtype=f)
    nxt = np.ascontiguousarray(next_humans_state, dtype=f)
    cur32 = np.ascontiguousarray(cur, dtype=f)
    rob = np.ascontiguousarray(current_robot_state, dtype=f)
    if single:
        nxt, cur32, rob = nxt[None], cur32[None], rob[None]
    W, n, A = cur32.shape[0], cur32.shape[1], acts.shape[0]
    oc = 15 if theta_and_omega_visible else 13
    if nxt.shape[-1] != (6 if theta_and_omega_visible else 4) or cur32.shape[-1] != (7 if theta_and_omega_visible else 5):
        raise ValueError("state column counts do not match theta_and_omega_visible")
    d_a, d_n, d_c, d_r = (DeviceBuffer.from_numpy(x) for x in (acts, nxt, cur32, rob))
    d_rot, d_rew = DeviceBuffer((W, A, n, oc)), DeviceBuffer((W, A))
    check(_lib.load().cs_lookahead(C.c_int(W), C.c_int(n), C.c_int(A), C.c_int(int(bool(theta_and_omega_visible))),
                                   C.c_void_p(d_a.ptr), C.c_void_p(d_n.ptr), C.c_void_p(d_c.ptr), C.c_void_p(d_r.ptr),
                                   C.c_int(rob.shape[-1]), C.c_float(dt), C.c_void_p(d_rot.ptr), C.c_void_p(d_rew.ptr), None))
    dtype = cur.dtype if cur.dtype in (np.float32, np.float64) else np.float64
    rot, rew = d_rot.download().astype(dtype), d_rew.download().astype(dtype)
    return (rot[0], rew[0]) if single else (rot, rew)


def compute_action_value(rewards, value_network_min_outputs, dt, gamma, vpref):
    """cadrl.py:85-90 (host numpy: A scalars)."""
    return np.asarray(rewards) + pow(gamma, dt * vpref) * np.asarray(value_network_min_outputs)


def propagate_humans_state_with_constant_velocity_model(current_humans_state, dt, theta_and_omega_visible=False):
    """cadrl.py:92-105 (host numpy)."""
    c = np.asarray(current_humans_state)
    if theta_and_omega_visible:
        return np.stack([c[:, 0] + c[:, 2] * dt, c[:, 1] + c[:, 3] * dt, c[:, 5] + c[:, 6] * dt, c[:, 2], c[:, 3], c[:, 6]], 1)
    return np.stack([c[:, 0] + c[:, 2] * dt, c[:, 1] + c[:, 3] * dt, c[:, 2], c[:, 3]], 1)
