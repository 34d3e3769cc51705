"""ctypes front-end of the C oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  It is the checker, never the product: the shipped package has no dependency on it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".inc"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _ptr(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64", C.c_double
    if dtype == np.float32:
        return "f32", C.c_float
    raise TypeError(dtype)


def update_humans(type_, S, goals, obstacles, P, dt, safety, all_params_equal=False, last_is_robot=False,
                  dtype=np.float64):
    """Same call shape as the reference's update_humans_parallel (forces_parallel.py:185).

    S and goals are converted to `dtype` copies; returns (out, S_after, goals_after)."""
    sfx, ct = _suffix(dtype)
    S = np.ascontiguousarray(S, dtype=dtype).copy()
    goals = np.ascontiguousarray(goals, dtype=dtype).copy()
    P = np.ascontiguousarray(P, dtype=dtype)
    safety = np.ascontiguousarray(safety, dtype=dtype)
    rows = S.shape[0]
    G = goals.shape[1]
    if obstacles is not None:
        obstacles = np.ascontiguousarray(obstacles, dtype=dtype)
        O, Smax = obstacles.shape[0], obstacles.shape[1]
    else:
        O, Smax = 0, 0
    out = np.empty_like(S)
    scratch = np.empty(rows * 2 + O * 2 + 8, dtype=dtype)
    fn = getattr(lib(), f"orc_update_humans_{sfx}")
    fn.restype = C.c_int
    rc = fn(C.c_int(type_), _ptr(S, ct), _ptr(goals, ct), C.c_int(G), _ptr(obstacles, ct), C.c_int(O),
            C.c_int(Smax), _ptr(P, ct), ct(dt), _ptr(safety, ct), C.c_int(int(all_params_equal)),
            C.c_int(int(last_is_robot)), C.c_int(rows), _ptr(out, ct), _ptr(scratch, ct))
    if rc != 0:
        raise ValueError(f"Type {type_} does not exist for this implementation")
    return out, S, goals


def step_block(type_, S, goals, obstacles, P, dt, n_substeps, safety, all_params_equal, robot_visible=False,
               robot=None, action=None, kinematics=0, respawn=False, respawn_par=(0.0, 0.0, 0.0),
               dtype=np.float64, threads=0):
    """Batched ([W, rows, 13] ...) or single-world ([rows, 13]) block of substeps; returns
    (S, goals, robot) after the block."""
    sfx, ct = _suffix(dtype)
    S = np.ascontiguousarray(S, dtype=dtype).copy()
    single = S.ndim == 2
    if single:
        S = S[None]
    W, rows = S.shape[0], S.shape[1]
    n = rows - int(robot_visible)
    goals = np.ascontiguousarray(goals, dtype=dtype).copy().reshape(W, n, -1, 2)
    G = goals.shape[2]
    P = np.ascontiguousarray(P, dtype=dtype)
    if P.ndim == 2:
        P = np.broadcast_to(P, (W,) + P.shape).copy()
    safety = np.ascontiguousarray(np.broadcast_to(np.asarray(safety, dtype=dtype), (W, rows)))
    obs_stride = 0
    if obstacles is not None:
        obstacles = np.ascontiguousarray(obstacles, dtype=dtype)
        if obstacles.ndim == 5:
            obs_stride = int(np.prod(obstacles.shape[1:]))
            O, Smax = obstacles.shape[1], obstacles.shape[2]
        else:
            O, Smax = obstacles.shape[0], obstacles.shape[1]
    else:
        O, Smax = 0, 0
    if robot is not None:
        robot = np.ascontiguousarray(robot, dtype=dtype).copy().reshape(W, 13)
    if action is not None:
        action = np.ascontiguousarray(np.broadcast_to(np.asarray(action, dtype=dtype), (W, 2)))
    rp = np.asarray(respawn_par, dtype=dtype)
    if threads <= 0:   # one OpenMP thread per world, at most what the cgroup grants
        threads = max(1, min(W, effective_cores()))
    fn = getattr(lib(), f"orc_step_block_batched_{sfx}")
    fn.restype = C.c_int
    rc = fn(C.c_int(W), C.c_int(type_), _ptr(S, ct), _ptr(goals, ct), C.c_int(G), _ptr(obstacles, ct),
            C.c_size_t(obs_stride), C.c_int(O), C.c_int(Smax), _ptr(P, ct), C.c_size_t(n * 20), ct(dt),
            C.c_int(n_substeps), _ptr(safety, ct), C.c_int(int(all_params_equal)), C.c_int(int(robot_visible)),
            C.c_int(rows), _ptr(robot, ct), _ptr(action, ct), C.c_int(kinematics), C.c_int(int(respawn)),
            _ptr(rp, ct), C.c_int(threads))
    if rc != 0:
        raise ValueError(f"oracle step_block failed rc={rc}")
    if single:
        return S[0], goals[0], (robot[0] if robot is not None else None)
    return S, goals, robot


class StepBlockRunner:
    """In-place, copy-free repeated calls of orc_step_block_batched (for timing the CPU baseline)."""

    def __init__(self, type_, S, goals, obstacles, P, safety, all_params_equal, respawn=False,
                 respawn_par=(0.0, 0.0, 0.0), dtype=np.float64, threads=0, robot=None, action=None):
        """``robot`` [W, 13] + ``action`` [W, 2]: a visible robot as the last state row of every world (the reference's robot_visible
        branch: the humans see it, its own row is moved by the action)."""
        self.sfx, self.ct = _suffix(dtype)
        self.S = np.ascontiguousarray(S, dtype=dtype).copy()
        self.W, self.rows = self.S.shape[0], self.S.shape[1]
        self.robot = None if robot is None else np.ascontiguousarray(robot, dtype=dtype).copy()
        self.action = None if action is None else np.ascontiguousarray(np.broadcast_to(np.asarray(action, dtype=dtype), (self.W, 2)))
        self.n = self.rows - (0 if self.robot is None else 1)
        self.goals = np.ascontiguousarray(goals, dtype=dtype).copy().reshape(self.W, self.n, -1, 2)
        self.G = self.goals.shape[2]
        P = np.ascontiguousarray(P, dtype=dtype)
        self.P = np.broadcast_to(P, (self.W,) + P.shape[-2:]).copy() if P.ndim == 2 else P
        self.safety = np.ascontiguousarray(np.broadcast_to(np.asarray(safety, dtype=dtype), (self.W, self.rows)))
        self.obs = None if obstacles is None else np.ascontiguousarray(obstacles, dtype=dtype)
        self.O, self.Smax = (0, 0) if self.obs is None else (self.obs.shape[-4], self.obs.shape[-3])
        self.obs_stride = 0 if (self.obs is None or self.obs.ndim == 4) else int(np.prod(self.obs.shape[1:]))
        self.type_, self.peq, self.respawn = type_, all_params_equal, respawn
        self.rp = np.asarray(respawn_par, dtype=dtype)
        self.threads = threads
        self.fn = getattr(lib(), f"orc_step_block_batched_{self.sfx}")
        self.fn.restype = C.c_int

    def run(self, dt, n_substeps):
        ct = self.ct
        rc = self.fn(C.c_int(self.W), C.c_int(self.type_), _ptr(self.S, ct), _ptr(self.goals, ct), C.c_int(self.G),
                     _ptr(self.obs, ct), C.c_size_t(self.obs_stride), C.c_int(self.O), C.c_int(self.Smax),
                     _ptr(self.P, ct), C.c_size_t(self.n * 20), ct(dt), C.c_int(n_substeps), _ptr(self.safety, ct),
                     C.c_int(int(self.peq)), C.c_int(0 if self.robot is None else 1), C.c_int(self.rows), _ptr(self.robot, ct), _ptr(self.action, ct), C.c_int(0),
                     C.c_int(int(self.respawn)), _ptr(self.rp, ct), C.c_int(self.threads))
        if rc != 0:
            raise ValueError(f"oracle step_block failed rc={rc}")


def respawn(S, goals, safety, robot, bound_x, bound_y, dtype=np.float64):
    sfx, ct = _suffix(dtype)
    S = np.ascontiguousarray(S, dtype=dtype).copy()
    goals = np.ascontiguousarray(goals, dtype=dtype).copy()
    safety = np.ascontiguousarray(safety, dtype=dtype)
    n, G = goals.shape[0], goals.shape[1]
    rb = None if robot is None else np.asarray(robot, dtype=dtype)
    fn = getattr(lib(), f"orc_respawn_{sfx}")
    fn.restype = None
    fn(_ptr(S, ct), _ptr(goals, ct), C.c_int(G), C.c_int(n), _ptr(safety, ct), _ptr(rb, ct), ct(bound_x), ct(bound_y))
    return S, goals


INFO_NAMES = ["Nothing", "Danger", "ReachGoal", "Collision", "Timeout"]


def collision_reward(hp, hv, hr, rp, rr, rgoal, act, T, global_time, time_limit=50.0, success_reward=1.0,
                     collision_penalty=-0.25, discomfort_dist=0.2, discomfort_factor=0.5, dtype=np.float64):
    sfx, ct = _suffix(dtype)
    hp = np.ascontiguousarray(hp, dtype=dtype); hv = np.ascontiguousarray(hv, dtype=dtype)
    hr = np.ascontiguousarray(hr, dtype=dtype); rp = np.ascontiguousarray(rp, dtype=dtype)
    rgoal = np.ascontiguousarray(rgoal, dtype=dtype); act = np.ascontiguousarray(act, dtype=dtype)
    out = np.zeros(7, dtype=dtype)
    fn = getattr(lib(), f"orc_collision_reward_{sfx}")
    fn.restype = None
    fn(C.c_int(len(hr)), _ptr(hp, ct), _ptr(hv, ct), _ptr(hr, ct), _ptr(rp, ct), ct(rr), _ptr(rgoal, ct),
       _ptr(act, ct), ct(T), ct(global_time), ct(time_limit), ct(success_reward), ct(collision_penalty),
       ct(discomfort_dist), ct(discomfort_factor), _ptr(out, ct))
    return dict(collision=bool(out[0]), dmin=float(out[1]), reaching_goal=bool(out[2]), reward=float(out[3]),
                terminated=bool(out[4]), truncated=bool(out[5]), info=INFO_NAMES[int(out[6])])


def lookahead(actions, nxt, cur, robot, dt, headed=False, dtype=np.float64):
    """compute_rotated_states_and_reward (cadrl.py:42-83): returns (rotated [A,n,13|15], rewards [A])."""
    sfx, ct = _suffix(dtype)
    actions = np.ascontiguousarray(actions, dtype=dtype); nxt = np.ascontiguousarray(nxt, dtype=dtype)
    cur = np.ascontiguousarray(cur, dtype=dtype); robot = np.ascontiguousarray(robot, dtype=dtype)
    A, n = actions.shape[0], cur.shape[0]
    rot = np.zeros((A, n, 15 if headed else 13), dtype=dtype)
    rew = np.zeros(A, dtype=dtype)
    fn = getattr(lib(), f"orc_lookahead_{sfx}")
    fn.restype = None
    fn(C.c_int(A), C.c_int(n), C.c_int(int(headed)), _ptr(actions, ct), _ptr(nxt, ct), _ptr(cur, ct), _ptr(robot, ct),
       ct(dt), _ptr(rot, ct), _ptr(rew, ct))
    return rot, rew


def process_obstacle(vertices) -> np.ndarray:
    """RVOSimulator::addObstacle for one polygon (vertex order as given, counter-clockwise): [n, 8] float32 records
    px, py, unitDir.x, unitDir.y, isConvex, next, prev, 0 with polygon-local next / prev indices."""
    v = np.asarray(vertices, dtype=np.float32).reshape(-1, 2)
    n = len(v)
    out = np.zeros((n, 8), np.float32)
    for i in range(n):
        nxt, prv = (i + 1) % n, (i - 1) % n
        d = v[nxt] - v[i]
        u = d / np.float32(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1])))
        if n == 2:
            convex = True
        else:  # leftOf(prev, this, next) >= 0, leftOf(a, b, c) = det(a - c, b - a)
            a, b, c = v[prv], v[i], v[nxt]
            convex = np.float32((a[0] - c[0]) * (b[1] - a[1])) - np.float32((a[1] - c[1]) * (b[0] - a[0])) >= 0
        out[i] = [v[i, 0], v[i, 1], u[0], u[1], 1.0 if convex else 0.0, nxt, prv, 0.0]
    return out


def split_obstacles(records) -> np.ndarray:
    """KdTree::buildObstacleTreeRecursive of RVO2 v2.0.x, the part that changes the obstacle set: every edge straddling the line of a node's
    splitting edge is cut at the intersection (new convex vertex with the cut edge's direction).  Written with an explicit work list (RVO2
    recurses left then right: the list is a stack that pops the left child first, so new vertices get RVO2's indices).  float32.  UNPINNED."""
    f = np.float32
    eps = f(1e-5)
    V = np.asarray(records, dtype=np.float32).reshape(-1, 8).copy()
    px, py = list(V[:, 0]), list(V[:, 1])
    ux, uy, cv = list(V[:, 2]), list(V[:, 3]), list(V[:, 4])
    nxt, prv = [int(x) for x in V[:, 5]], [int(x) for x in V[:, 6]]

    def side(i1, i2, j):          # leftOf(point[i1], point[i2], point[j]) = det(p1 - pj, p2 - p1)
        return f(f(f(px[i1] - px[j]) * f(py[i2] - py[i1])) - f(f(py[i1] - py[j]) * f(px[i2] - px[i1])))

    stack = [list(range(len(px)))]
    while stack:
        obst = stack.pop()
        m = len(obst)
        if m == 0:
            continue
        best_key, best = (m, m), 0
        for i, i1 in enumerate(obst):
            i2 = nxt[i1]
            l = r = 0
            for j, j1 in enumerate(obst):
                if j == i:
                    continue
                a, b = side(i1, i2, j1), side(i1, i2, nxt[j1])
                if a >= -eps and b >= -eps:
                    l += 1
                elif a <= eps and b <= eps:
                    r += 1
                else:
                    l += 1; r += 1
            key = (max(l, r), min(l, r))
            if key < best_key:
                best_key, best = key, i
        i1 = obst[best]; i2 = nxt[i1]
        L, R = [], []
        for j, j1 in enumerate(obst):
            if j == best:
                continue
            j2 = nxt[j1]
            a, b = side(i1, i2, j1), side(i1, i2, j2)
            if a >= -eps and b >= -eps:
                L.append(j1)
            elif a <= eps and b <= eps:
                R.append(j1)
            else:
                ex, ey = f(px[i2] - px[i1]), f(py[i2] - py[i1])
                num = f(f(ex * f(py[j1] - py[i1])) - f(ey * f(px[j1] - px[i1])))
                den = f(f(ex * f(py[j1] - py[j2])) - f(ey * f(px[j1] - px[j2])))
                t = f(num / den)
                k = len(px)
                px.append(f(px[j1] + f(t * f(px[j2] - px[j1])))); py.append(f(py[j1] + f(t * f(py[j2] - py[j1]))))
                ux.append(ux[j1]); uy.append(uy[j1]); cv.append(f(1.0)); nxt.append(j2); prv.append(j1)
                nxt[j1] = k; prv[j2] = k
                (L if a > 0 else R).append(j1)
                (R if a > 0 else L).append(k)
        stack.append(R)           # popped second
        stack.append(L)           # popped first: RVO2 builds the left subtree (and numbers its new vertices) before the right one
    out = np.zeros((len(px), 8), np.float32)
    out[:, 0], out[:, 1], out[:, 2], out[:, 3], out[:, 4], out[:, 5], out[:, 6] = px, py, ux, uy, cv, nxt, prv
    return out


def process_obstacles(polygons, kdtree_split=True) -> np.ndarray:
    """All polygons of a scene in one vertex table (next / prev become global indices), then processObstacles()' edge splitting."""
    recs, base = [], 0
    for poly in polygons:
        r = process_obstacle(poly)
        r[:, 5] += base
        r[:, 6] += base
        base += len(r)
        recs.append(r)
    out = np.concatenate(recs) if recs else np.zeros((0, 8), np.float32)
    return split_obstacles(out) if (kdtree_split and len(out)) else out


def orca_new_velocities_obst(pos, vel, pref, radius, maxspeed, verts, neighbor_dist=10.0, max_nb=10, time_horizon=5.0,
                             time_horizon_obst=5.0, time_step=0.25):
    """One RVO2 doStep velocity solve with static obstacles (float32, PARITY UNPINNED).
    Returns (new_vel [na, 2], lines [na, max_nb + 32, 4], nlines [na], nobst [na]); obstacle lines come first."""
    f = np.float32
    pos = np.ascontiguousarray(pos, dtype=f); vel = np.ascontiguousarray(vel, dtype=f)
    pref = np.ascontiguousarray(pref, dtype=f); radius = np.ascontiguousarray(radius, dtype=f)
    maxspeed = np.ascontiguousarray(maxspeed, dtype=f)
    verts = np.ascontiguousarray(verts, dtype=f).reshape(-1, 8)
    na = pos.shape[0]
    out = np.zeros((na, 2), dtype=f)
    lines = np.zeros((na, max_nb + 32, 4), dtype=f)
    nl = np.zeros(na, dtype=np.int32)
    no = np.zeros(na, dtype=np.int32)
    fn = lib().orc_orca_new_velocities_obst
    fn.restype = None
    fn(C.c_int(na), _ptr(pos, C.c_float), _ptr(vel, C.c_float), _ptr(pref, C.c_float), _ptr(radius, C.c_float),
       _ptr(maxspeed, C.c_float), C.c_float(neighbor_dist), C.c_int(max_nb), C.c_float(time_horizon),
       C.c_float(time_horizon_obst), C.c_float(time_step), _ptr(verts, C.c_float), C.c_int(len(verts)), _ptr(out, C.c_float),
       lines.ctypes.data_as(C.c_void_p), nl.ctypes.data_as(C.POINTER(C.c_int)), no.ctypes.data_as(C.POINTER(C.c_int)))
    return out, lines, nl, no


def orca_new_velocities(pos, vel, pref, radius, maxspeed, neighbor_dist=10.0, max_nb=10, time_horizon=5.0,
                        time_step=0.25, return_lines=False):
    """One RVO2 doStep velocity solve for one world (float32, PARITY UNPINNED)."""
    f = np.float32
    pos = np.ascontiguousarray(pos, dtype=f); vel = np.ascontiguousarray(vel, dtype=f)
    pref = np.ascontiguousarray(pref, dtype=f); radius = np.ascontiguousarray(radius, dtype=f)
    maxspeed = np.ascontiguousarray(maxspeed, dtype=f)
    na = pos.shape[0]
    out = np.zeros((na, 2), dtype=f)
    lines = np.zeros((na, max(max_nb, 1), 4), dtype=f)
    nl = np.zeros(na, dtype=np.int32)
    fn = lib().orc_orca_new_velocities
    fn.restype = None
    fn(C.c_int(na), _ptr(pos, C.c_float), _ptr(vel, C.c_float), _ptr(pref, C.c_float), _ptr(radius, C.c_float),
       _ptr(maxspeed, C.c_float), C.c_float(neighbor_dist), C.c_int(max_nb), C.c_float(time_horizon),
       C.c_float(time_step), _ptr(out, C.c_float), lines.ctypes.data_as(C.c_void_p) if return_lines else None,
       nl.ctypes.data_as(C.POINTER(C.c_int)) if return_lines else None)
    return (out, lines, nl) if return_lines else out


def orca_step_block(S, goals, margin, dt, n_substeps, robot_visible=False, robot=None, action=None,
                    neighbor_dist=10.0, max_nb=10, time_horizon=5.0, respawn=False, bounds=(0.0, 0.0), threads=0,
                    verts=None, time_horizon_obst=5.0, agent_params=None):
    """Batched ([W, rows, 13]) or single-world block of ORCA substeps on the shared row layout
    (cols 5:7 = preferred velocity).  Returns (S, goals, robot).  ``agent_params`` [W, rows, 4] (or [rows, 4]): RVO2's per-agent
    neighborDist, maxNeighbors, timeHorizon, timeHorizonObst (``max_nb`` must then be the largest maxNeighbors)."""
    f = np.float32
    S = np.ascontiguousarray(S, dtype=f).copy()
    single = S.ndim == 2
    if single:
        S = S[None]
    W, rows = S.shape[0], S.shape[1]
    n = rows - int(robot_visible)
    goals = np.ascontiguousarray(goals, dtype=f).copy().reshape(W, n, -1, 2)
    G = goals.shape[2]
    margin = np.ascontiguousarray(np.broadcast_to(np.asarray(margin, dtype=f), (W, rows)))
    if robot is not None:
        robot = np.ascontiguousarray(robot, dtype=f).copy().reshape(W, 13)
    if action is not None:
        action = np.ascontiguousarray(np.broadcast_to(np.asarray(action, dtype=f), (W, 2)))
    vt = None if verts is None else np.ascontiguousarray(verts, dtype=f).reshape(-1, 8)
    ap = None if agent_params is None else np.ascontiguousarray(np.broadcast_to(np.asarray(agent_params, dtype=f), (W, rows, 4)))
    fn = lib().orc_orca_step_block_batched_pa
    fn.restype = None
    threads = int(threads) or effective_cores()   # (an OpenMP team of the box's 256 logical CPUs under a 16-CPU quota costs ~0.1 s per call)
    fn(C.c_int(W), _ptr(S, C.c_float), _ptr(goals, C.c_float), C.c_int(G), C.c_int(rows), C.c_int(int(robot_visible)),
       _ptr(margin, C.c_float), _ptr(robot, C.c_float), _ptr(action, C.c_float), C.c_float(dt), C.c_int(n_substeps),
       C.c_float(neighbor_dist), C.c_int(max_nb), C.c_float(time_horizon), C.c_int(int(respawn)),
       C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_int(threads), C.c_float(time_horizon_obst), _ptr(vt, C.c_float),
       C.c_int(0 if vt is None else len(vt)), _ptr(ap, C.c_float))
    if single:
        return S[0], goals[0], (robot[0] if robot is not None else None)
    return S, goals, robot


def orca_step_block_f64(S, goals, margin, dt, n_substeps, neighbor_dist=10.0, max_nb=10, time_horizon=5.0, threads=0):
    """orca_step_block with the restatement instantiated in DOUBLE (oracle/orca_oracle_f64.c): the algorithm without float32
    rounding, from the same (float32-valued) rows.  A classification aid for tests/orca_fast_parity.py, not a second reference:
    RVO2 itself is float32.  Plain crowd batches only (no robot, no obstacles, no respawn).  Returns (S, goals) as float64."""
    S = np.ascontiguousarray(S, dtype=np.float64).copy()
    single = S.ndim == 2
    if single:
        S = S[None]
    W, rows = S.shape[0], S.shape[1]
    goals = np.ascontiguousarray(goals, dtype=np.float64).copy().reshape(W, rows, -1, 2)
    G = goals.shape[2]
    margin = np.ascontiguousarray(np.broadcast_to(np.asarray(margin, dtype=np.float64), (W, rows)))
    fn = lib().orc64_orca_step_block_batched_pa
    fn.restype = None
    threads = int(threads) or effective_cores()
    fn(C.c_int(W), _ptr(S, C.c_double), _ptr(goals, C.c_double), C.c_int(G), C.c_int(rows), C.c_int(0), _ptr(margin, C.c_double), None, None,
       C.c_double(dt), C.c_int(n_substeps), C.c_double(neighbor_dist), C.c_int(max_nb), C.c_double(time_horizon), C.c_int(0),
       C.c_double(0.0), C.c_double(0.0), C.c_int(threads), C.c_double(5.0), None, C.c_int(0), None)
    return (S[0], goals[0]) if single else (S, goals)


ORCA_DECISION_KINDS = ["", "neighbour in range", "neighbour order", "colliding or not", "cut-off circle: side", "cut-off circle: cone", "which leg",
                       "LP2: preferred velocity clipped", "LP2: line violated", "LP1: line misses the speed circle", "LP1: parallel lines",
                       "LP1: parallel lines, side", "LP1: sign of the denominator", "LP1: empty interval", "LP1: direction optimum", "LP1: clamp left",
                       "LP1: clamp right", "LP3: line violated", "LP3: parallel lines", "LP3: same direction", "LP2 failed -> LP3", "LP3: inner LP2 failed"]


def orca_probe_agent(S_w, margin_w, agent, dt, k_ulps=4.0, probes=65, seed=1, neighbor_dist=10.0, max_nb=10, time_horizon=5.0, cap=4096):
    """The PROBE instantiation of the ORCA restatement (oracle/orca_oracle_probe.c) for ONE agent of ONE world given as rows [n, 13]
    (+ margins [n]): its new velocity `probes` times -- probe 0 unperturbed (== orca_step_block's velocity, bit for bit), the others with
    every division / square root / two-term product sum perturbed by <= k_ulps float32 ulps -- and the ordered decision trace of each.
    Returns (vel [probes, 2] float32, traces: list of (kind [m], outcome [m], margin [m]) arrays)."""
    f = np.float32
    S_w = np.ascontiguousarray(S_w, dtype=f)
    n = S_w.shape[0]
    pos = np.ascontiguousarray(S_w[:, 0:2]); vel = np.ascontiguousarray(S_w[:, 3:5]); pref = np.ascontiguousarray(S_w[:, 5:7])
    rad = np.ascontiguousarray(S_w[:, 8] + np.asarray(margin_w, dtype=f)); vmax = np.ascontiguousarray(S_w[:, 12])
    out = np.zeros((probes, 2), f)
    kind = np.zeros((probes, cap), np.int32); outc = np.zeros((probes, cap), np.int32); marg = np.zeros((probes, cap), f)
    nt = np.zeros(probes, np.int32)
    fn = lib().orcp_probe_agent
    fn.restype = None
    fn(C.c_int(n), _ptr(pos, C.c_float), _ptr(vel, C.c_float), _ptr(pref, C.c_float), _ptr(rad, C.c_float), _ptr(vmax, C.c_float),
       C.c_float(neighbor_dist), C.c_int(max_nb), C.c_float(time_horizon), C.c_float(dt), C.c_int(int(agent)), C.c_uint64(int(seed)), C.c_float(k_ulps),
       C.c_int(probes), _ptr(out, C.c_float), _ptr(kind, C.c_int), _ptr(outc, C.c_int), _ptr(marg, C.c_float), C.c_int(cap), _ptr(nt, C.c_int))
    traces = [(kind[p, :min(nt[p], cap)], outc[p, :min(nt[p], cap)], marg[p, :min(nt[p], cap)]) for p in range(probes)]
    return out, traces


def effective_cores() -> int:
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box shows 256 logical CPUs
    and grants 16 through cpu.max: an OpenMP team of 256 only gets throttled, and costs ~0.1 s to spin up per call)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def num_threads() -> int:
    f = lib().orc_num_threads
    f.restype = C.c_int
    return int(f())


def laser_scan(pos, yaw, rng, samples, max_distance, human_pos, human_radius, obstacles=None, dtype=np.float64):
    """numpy restatement of LaserSensor.get_laser_measurements without noise
    (/root/reference/social_gym/src/sensors.py:51-66; disc hit :24-33, one-sided segment hit :35-49).
    obstacles: [O][Smax][2][2] NaN-padded, endpoints ordered as Obstacle.segments stores them (obstacle.py:31-32).
    Returns (angles [samples], measurements [samples])."""
    dt = np.dtype(dtype).type
    pos = np.asarray(pos, dtype)
    angles = np.linspace(dt(yaw) - dt(rng) / dt(2), dt(yaw) + dt(rng) / dt(2), int(samples), dtype=dtype)   # :53
    md = dt(max_distance)
    out = np.full(int(samples), md, dtype)
    dirs = np.stack([np.cos(angles), np.sin(angles)], axis=-1).astype(dtype)                                   # :57
    hp = np.asarray(human_pos, dtype).reshape(-1, 2)
    hr = np.asarray(human_radius, dtype).reshape(-1)
    for k in range(int(samples)):
        d = dirs[k]
        m = md
        for i in range(len(hp)):                                                                             # :58-60
            s = pos - hp[i]
            b = s[0] * d[0] + s[1] * d[1]
            c = s[0] * s[0] + s[1] * s[1] - hr[i] * hr[i]
            h = b * b - c
            if h < 0:
                continue
            t = -b - np.sqrt(h)
            if t < 0:
                continue
            m = min(m, min(t, md))
        if obstacles is not None:
            segs = np.asarray(obstacles, dtype).reshape(-1, 2, 2)
            for sg in segs:                                                                                  # :61-64
                if np.isnan(sg[0, 0]):
                    continue
                x1, y1, x2, y2 = sg[0, 0], sg[0, 1], sg[1, 0], sg[1, 1]
                x3, y3 = pos[0], pos[1]
                x4, y4 = pos[0] + d[0], pos[1] + d[1]
                den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
                if den <= 0:
                    continue
                t = ((x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)) / den
                u = -((x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)) / den
                if 0 < t < 1 and u > 0:
                    ix, iy = x1 + t * (x2 - x1), y1 + t * (y2 - y1)
                    m = min(m, min(np.sqrt((x3 - ix) ** 2 + (y3 - iy) ** 2), md))
        out[k] = m
    return angles, out


def social_momentum_step(pos, vel, radius, safety, vd, goals, dt, robot=None, n_actions=20, lam=0.11, dtype=np.float64,
                         amb_eps=None):
    """numpy restatement of one update_humans(t, dt) of the social-momentum crowd model
    (/root/reference/social_gym/src/motion_model_manager.py:395-404 with :66-70 update_goals and :247-251 action set;
    /root/reference/social_gym/src/social_momentum.py:10-27 filter, :29-44 reactive agents, :46-75 optimize_momentum).
    robot = [px, py, vx, vy, radius, safety] when the robot is considered, else None.
    Returns (pos', vel', goals', rewards [n, A] with -inf for filtered actions, chosen [n] action index or -1).

    The model tests the SIGN of `current_momentum * expected_momentum` (:69).  That product is exactly zero in exact
    arithmetic whenever two agents move with the same velocity, or a candidate action equals the partner's velocity --
    both common with 20 discrete headings and equal speeds -- and the reference then follows float64 rounding noise.
    With `amb_eps` the function also returns, as a 6th value, (lo, hi) [n, A]: the smallest / largest reward each action
    can get when every pair with |product| < amb_eps may fall either way (a zeroed momentum term or the full sum)."""
    pos = np.array(pos, dtype); vel = np.array(vel, dtype); goals = np.array(goals, dtype)
    radius = np.asarray(radius, dtype); safety = np.asarray(safety, dtype); vd = np.asarray(vd, dtype)
    n = len(pos)
    dt = np.dtype(dtype).type(dt)
    ang = np.array([((2 * np.pi) / n_actions) * i for i in range(n_actions)])                        # :249-250
    unit = np.stack([np.cos(ang), np.sin(ang)], axis=-1).astype(dtype)
    ep, ev = pos, vel
    er, es = radius, safety
    if robot is not None and len(robot):                                                                  # entities (:13-14)
        rb = np.asarray(robot, dtype)
        ep = np.vstack([pos, rb[None, 0:2]]); ev = np.vstack([vel, rb[None, 2:4]])
        er = np.append(radius, rb[4]); es = np.append(safety, rb[5])
    rewards = np.full((n, n_actions), -np.inf, dtype)
    r_lo = np.full((n, n_actions), -np.inf, dtype)
    r_hi = np.full((n, n_actions), -np.inf, dtype)
    chosen = np.full(n, -1)
    new_vel = np.zeros_like(vel)
    for i in range(n):
        gi = goals[i]
        k = int(np.argmax(np.isnan(gi[:, 0]))) if np.isnan(gi[:, 0]).any() else len(gi)
        if k and np.linalg.norm(gi[0] - pos[i]) < radius[i]:                                             # update_goals :66-70
            goals[i, :k] = np.roll(gi[:k], -1, axis=0)
        others = np.array([j for j in range(len(ep)) if j != i], dtype=int)
        acts = unit * vd[i]
        nxt_i = pos[i] + acts * dt                                                                       # [A, 2]
        nxt_o = ep[others] + ev[others] * dt
        dist = np.linalg.norm(nxt_o[None] - nxt_i[:, None], axis=-1)                                     # [A, J]
        thr = radius[i] + er[others] + safety[i] + es[others]
        free = ~np.any(dist < thr[None], axis=1) if len(others) else np.ones(n_actions, bool)           # :16-26
        diff = ep[others] - pos[i]
        dn = np.linalg.norm(diff, axis=-1)
        speed = np.linalg.norm(vel[i])
        with np.errstate(invalid="ignore", divide="ignore"):
            if speed == 0:
                angle = np.zeros(len(others), dtype)
            else:
                angle = np.arccos(np.clip(diff @ vel[i] / (speed * dn), -1, 1))
            react = others[angle <= np.pi]                                                               # :36-43 (NaN -> not reactive)
            w = 1 / np.linalg.norm(ep[react] - pos[i], axis=-1)
            w = w / np.sum(w)                                                                            # :49-52
        best, best_a = -100000, -1
        for a in range(n_actions):
            if not free[a]:
                continue
            nd = np.linalg.norm(goals[i, 0] - (pos[i] + acts[a] * dt))
            rew = 1 / nd
            mom = 0
            mom_full, amb, hard_break = 0, False, False
            for t, j in enumerate(react):
                c = (pos[i] + ep[j]) / 2
                pr, ph = pos[i] - c, ep[j] - c
                other = ph[0] * ev[j][1] - ph[1] * ev[j][0]
                cur = pr[0] * vel[i][1] - pr[1] * vel[i][0] + other
                exp = pr[0] * acts[a][1] - pr[1] * acts[a][0] + other
                if amb_eps is not None and not hard_break:
                    if abs(cur * exp) < amb_eps:
                        amb = True                      # either way: contributes ~0 if kept
                    elif cur * exp > 0:
                        mom_full += w[t] * exp
                    else:
                        hard_break = True
            for t, j in enumerate(react):
                c = (pos[i] + ep[j]) / 2
                pr, ph = pos[i] - c, ep[j] - c
                other = ph[0] * ev[j][1] - ph[1] * ev[j][0]
                cur = pr[0] * vel[i][1] - pr[1] * vel[i][0] + other
                exp = pr[0] * acts[a][1] - pr[1] * acts[a][0] + other
                if cur * exp > 0:
                    mom += w[t] * exp
                else:
                    mom = 0
                    break
            base_rew = rew
            rew = rew + lam * mom
            rewards[i, a] = rew
            if amb_eps is not None:
                full = base_rew if hard_break else base_rew + lam * mom_full
                cands = [full, base_rew] if amb else [full]
                r_lo[i, a], r_hi[i, a] = min(cands), max(cands)
            if rew > best:
                best, best_a = rew, a
        chosen[i] = best_a
        if best_a >= 0:
            new_vel[i] = acts[best_a]
    new_pos = pos + vel * dt                                                                             # :401-403
    if amb_eps is not None:
        return new_pos, new_vel, goals, rewards, chosen, (r_lo, r_hi)
    return new_pos, new_vel, goals, rewards, chosen


# ---------------------------------------------------------------------------------------------------------------------
# The robot driven by a human motion model (imitation learning).  Restates motion_model_manager.py:552-653
# (set_robot_motion_model / compute_robot_forces / update_robot) with the single-agent force functions of forces.py
# (:9-16 desired, :27-53 obstacle Helbing / Guo, :153-218 social Helbing / Guo / Moussaid, :279-290 torque) and the Euler
# updates of motion_model_manager.py:72-86.  float64 like the reference.  Pinned on tests/golden/g11_imitation.npz.
# The single-agent functions are NOT the parallel ones: Guo's obstacle force is a plain sum (no mean), a robot within its
# radius of the goal keeps the desired force of the previous substep, the social force sums over the humans in index order.
ROBOT_MODELS = ["sfm_helbing", "sfm_guo", "sfm_moussaid", "hsfm_farina", "hsfm_guo", "hsfm_moussaid", "hsfm_new",
                "hsfm_new_guo", "hsfm_new_moussaid"]
# parameter slots (agent.py:269): 0 relaxation_time 1 Ai 2 Aw 3 Bi 4 Bw 5 Ci 6 Cw 7 Di 8 Dw 9 Ei 10 k1 11 k2 12 lambda
# 13 gamma 14 ns 15 ns1 16 ko 17 kd 18 alpha 19 k_lambda


def _bound_angle(a):
    """utils.py:7-13 (keeps +pi and -pi as they are)."""
    import math
    two_pi = 2 * math.pi
    if a >= two_pi:
        a = math.fmod(a, two_pi)
    if a <= -two_pi:
        a = math.fmod(a, two_pi)     # Python's a % -two_pi of a negative a
    if a > math.pi:
        a -= two_pi
    if a < -math.pi:
        a += two_pi
    return a


def closest_points(walls, p):
    """Wall.get_closest_point for every wall (obstacle.py:53-66): the LAST nearest point over the segments (<=)."""
    out = []
    for wall in ([] if walls is None else walls):
        best, bd = np.zeros(2), 10000.0
        for seg in wall:
            if np.any(np.isnan(seg)):
                continue
            a, b = seg[0], seg[1]
            t = np.dot(p - a, b - a) / (np.linalg.norm(b - a) ** 2)
            h = a + min(max(0.0, t), 1.0) * (b - a)
            d = np.linalg.norm(h - p)
            if d <= bd:
                best, bd = h, d
        out.append(best)
    return out


def _robot_forces(row, P, model, hum_pos, hum_vel, hum_radius, hum_safety, walls):
    """compute_robot_forces (motion_model_manager.py:591-613) on a robot row (see robot_model_substep): returns
    (vel, R, fd, fo, fs) -- the linear velocity (refreshed from the body velocity for headed models), the rotation matrix of the
    yaw, and the desired (kept within one radius of the goal), obstacle and social forces."""
    r = np.array(row, dtype=np.float64)
    headed = model.startswith("hsfm")
    guo, mou = model.endswith("guo"), model.endswith("moussaid")
    pos, yaw, vel, bvel, om = r[0:2].copy(), r[2], r[3:5].copy(), r[5:7].copy(), r[7]
    radius, mass, goal, vd, safety = r[8], r[9], r[10:12], r[12], r[13]
    inertia = 0.5 * mass * radius * radius
    R = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
    if headed:
        vel = R @ bvel
    # desired force: kept from the previous substep when within one radius of the goal (forces.py:12-16)
    diff = goal - pos
    dist = np.linalg.norm(diff)
    fd = r[14:16].copy()
    if dist > radius:
        fd = mass * (diff / dist * vd - vel) / P[0]
    # obstacle force
    fo = np.zeros(2)
    obs = closest_points(walls, pos)
    for o in obs:
        d = pos - o
        dn = np.linalg.norm(d)
        n_iw = d / dn
        t_iw = np.array([-n_iw[1], n_iw[0]])
        dv = -np.dot(vel, t_iw)
        rd = radius + safety - dn
        if guo:
            fo += (P[2] * np.exp(rd / P[4]) + P[10] * max(0, rd)) * n_iw + (-P[6] * np.exp(rd / P[8]) - P[11] * max(0, rd)) * dv * t_iw
        else:
            fo += (P[2] * np.exp(rd / P[4]) + P[10] * max(0, rd)) * n_iw - P[11] * max(0, rd) * dv * t_iw
    if obs and not guo:
        fo /= len(obs)
    # social force from the humans, index order
    fs = np.zeros(2)
    for j in range(len(hum_pos)):
        rij = radius + safety + hum_radius[j] + hum_safety[j]
        d = pos - hum_pos[j]
        dn = np.linalg.norm(d)
        n_ij = d / dn
        rd = rij - dn
        if mou:
            iv = P[12] * (vel - hum_vel[j]) - n_ij
            inorm = np.linalg.norm(iv)
            i_ij = iv / inorm
            th = _bound_angle(np.arctan2(n_ij[1], n_ij[0]) - np.arctan2(i_ij[1], i_ij[0]) + np.pi)
            k = np.sign(th)
            h_ij = np.array([-i_ij[1], i_ij[0]])
            F = P[13] * inorm
            dvh = np.dot(hum_vel[j] - vel, h_ij)
            fs -= (P[9] * np.exp(-dn / F) * (np.exp(-(P[15] * F * th) ** 2) * i_ij + k * np.exp(-(P[14] * F * th) ** 2) * h_ij)
                   + P[10] * max(0, rd) * i_ij + P[11] * max(0, rd) * dvh * h_ij)
        else:
            t_ij = np.array([-n_ij[1], n_ij[0]])
            dv = np.dot(hum_vel[j] - vel, t_ij)
            if guo:
                fs += (P[1] * np.exp(rd / P[3]) + P[10] * max(0, rd)) * n_ij + (P[5] * np.exp(rd / P[7]) + P[11] * max(0, rd) * dv) * t_ij
            else:
                fs += (P[1] * np.exp(rd / P[3]) + P[10] * max(0, rd)) * n_ij + P[11] * max(0, rd) * dv * t_ij
    return vel, R, fd, fo, fs


def robot_model_substep(row, P, model, hum_pos, hum_vel, hum_radius, hum_safety, walls, dt, just_velocities=False):
    """One update_robot(t, dt) of an SFM / HSFM robot.  row = [x, y, yaw, vx, vy, bvx, bvy, omega, radius, mass, gx, gy,
    desired_speed, safety_space, desired_force_x, desired_force_y] (copied); returns the new row."""
    r = np.array(row, dtype=np.float64)
    headed = model.startswith("hsfm")
    pos, yaw, bvel, om = r[0:2].copy(), r[2], r[5:7].copy(), r[7]
    radius, mass, vd = r[8], r[9], r[12]
    inertia = 0.5 * mass * radius * radius
    vel, R, fd, fo, fs = _robot_forces(r, P, model, hum_pos, hum_vel, hum_radius, hum_safety, walls)
    if not headed:
        gf = fd + fo + fs
        if not just_velocities:                      # motion_model_manager.py:73
            pos = pos + vel * dt
        vel = vel + gf / mass * dt
        sp = np.linalg.norm(vel)
        if sp > vd:
            vel = vel / sp * vd
    else:
        tot = fd if model in ("hsfm_farina", "hsfm_guo", "hsfm_moussaid") else fd + fo + fs
        tn = np.linalg.norm(tot)
        k_theta = inertia * P[19] * tn
        k_omega = inertia * (1 + P[18]) * np.sqrt(P[19] * tn / P[18])
        torque = -k_theta * _bound_angle(yaw - np.arctan2(tot[1], tot[0])) - k_omega * om
        gf = np.array([np.dot(fd + fo + fs, R[:, 0]), P[16] * np.dot(fo + fs, R[:, 1]) - P[17] * bvel[1]])
        if not just_velocities:                      # motion_model_manager.py:79-81
            pos = pos + vel * dt
            yaw = _bound_angle(yaw + om * dt)
        bvel = bvel + gf / mass * dt
        om = om + torque / inertia * dt
        sp = np.linalg.norm(bvel)
        if sp > vd:
            bvel = bvel / sp * vd
        vel = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]) @ bvel
    r[0:2], r[2], r[3:5], r[5:7], r[7], r[14:16] = pos, yaw, vel, bvel, om, fd
    return r


# ---------------------------------------------------------------------------------------------------------------------
# RK45 integration of the crowd: MotionModelManager(runge_kutta=True).update_humans (motion_model_manager.py:374-384) =
# scipy.integrate.solve_ivp(method="RK45", default rtol 1e-3 / atol 1e-6) around f_rk45_headed / f_rk45_not_headed (:500-550).
# The right-hand side is NOT a pure function: it writes the trial state into the agents with the speed clamp and the angle
# wrap (:88-103), rotates goal lists at trial positions (compute_forces :439-441), and uses the single-agent force functions
# (stale desired force, Guo's un-averaged obstacle force; pairwise-mirrored social forces when all parameters are equal,
# forces.py:130-151).  scipy is the reference's own dependency (1.15.3 in this image): the oracle calls it, with the
# right-hand side restated here.  Pinned on tests/golden/g12_rk45.npz (rows and number of RHS evaluations).
def _pair_force(kind, P, pi, vi, ri, pj, vj, rj):
    """compute_pairwise_social_force(type, agent1 = i, agent2 = j) (forces.py:63-128); ri / rj = radius + safety_space."""
    d = pi - pj
    dn = np.linalg.norm(d)
    n_ij = d / dn
    rd = ri + rj - dn
    if kind == 2:
        iv = P[12] * (vi - vj) - n_ij
        inorm = np.linalg.norm(iv)
        i_ij = iv / inorm
        th = _bound_angle(np.arctan2(n_ij[1], n_ij[0]) - np.arctan2(i_ij[1], i_ij[0]) + np.pi)
        k = np.sign(th)
        h_ij = np.array([-i_ij[1], i_ij[0]])
        F = P[13] * inorm
        dvh = np.dot(vj - vi, h_ij)
        return -(P[9] * np.exp(-dn / F) * (np.exp(-(P[15] * F * th) ** 2) * i_ij + k * np.exp(-(P[14] * F * th) ** 2) * h_ij)
                 + P[10] * max(0, rd) * i_ij + P[11] * max(0, rd) * dvh * h_ij)
    t_ij = np.array([-n_ij[1], n_ij[0]])
    dv = np.dot(vj - vi, t_ij)
    if kind == 1:
        return (P[1] * np.exp(rd / P[3]) + P[10] * max(0, rd)) * n_ij + (P[5] * np.exp(rd / P[7]) + P[11] * max(0, rd) * dv) * t_ij
    return (P[1] * np.exp(rd / P[3]) + P[10] * max(0, rd)) * n_ij + P[11] * max(0, rd) * dv * t_ij


class Rk45Crowd:
    """State of one world between calls: rows [n, 16] = x, y, yaw, vx, vy, bvx, bvy, omega, radius, mass, gx, gy, desired_speed,
    safety_space, desired_force_x, desired_force_y; goals [n, G, 2] NaN-padded lists; params [n, 20]."""

    def __init__(self, rows, goals, params, model, all_equal, walls=None, robot=None):
        self.rows = np.array(rows, dtype=np.float64)
        self.goals = [[g.copy() for g in gl if not np.any(np.isnan(g))] for gl in np.asarray(goals, dtype=np.float64)]
        self.P = np.asarray(params, dtype=np.float64)
        self.model, self.all_equal = model, bool(all_equal)
        self.headed = model.startswith("hsfm")
        self.kind = 1 if model.endswith("guo") else (2 if model.endswith("moussaid") else 0)
        self.torque_new = model.startswith("hsfm_new")
        self.walls = walls if (walls is not None and len(walls)) else None
        self.robot = None if robot is None or len(robot) == 0 else np.asarray(robot, dtype=np.float64)  # x, y, vx, vy, radius, safety
        self.nfev = 0

    def _set_state(self, y):
        r = self.rows
        n = len(r)
        if self.headed:
            Y = y.reshape(n, 6)
            r[:, 0:2] = Y[:, 0:2]
            for i in range(n):
                r[i, 2] = _bound_angle(Y[i, 2])
                bv = Y[i, 3:5]
                sp = np.linalg.norm(bv)
                r[i, 5:7] = bv / sp * r[i, 12] if sp > r[i, 12] else bv
            r[:, 7] = Y[:, 5]
        else:
            Y = y.reshape(n, 4)
            r[:, 0:2] = Y[:, 0:2]
            for i in range(n):
                v = Y[i, 2:4]
                sp = np.linalg.norm(v)
                r[i, 3:5] = v / sp * r[i, 12] if sp > r[i, 12] else v

    def _forces(self):
        r = self.rows
        n = len(r)
        obstacles = []
        for i in range(n):
            gl = self.goals[i]
            if gl and np.linalg.norm(gl[0] - r[i, 0:2]) < r[i, 8]:   # update_goals (:66-70)
                gl.append(gl.pop(0))
            obstacles.append(closest_points(self.walls, r[i, 0:2]))
            if self.headed:
                c, s = np.cos(r[i, 2]), np.sin(r[i, 2])
                r[i, 3:5] = np.array([[c, -s], [s, c]]) @ r[i, 5:7]
        rs = r[:, 8] + r[:, 13]
        ent_p = [r[i, 0:2] for i in range(n)]
        ent_v = [r[i, 3:5] for i in range(n)]
        ent_r = list(rs)
        if self.robot is not None:
            ent_p.append(self.robot[0:2]); ent_v.append(self.robot[2:4]); ent_r.append(self.robot[4] + self.robot[5])
        fs = np.zeros((n, 2))
        if self.all_equal:   # compute_all_social_forces: the pair force of the lower index, mirrored onto the higher one
            for i in range(n):
                for j in range(i + 1, len(ent_p)):
                    f = _pair_force(self.kind, self.P[i], ent_p[i], ent_v[i], ent_r[i], ent_p[j], ent_v[j], ent_r[j])
                    fs[i] += f
                    if j != n:
                        fs[j] -= f
        else:
            for i in range(n):
                for j in range(len(ent_p)):
                    if j != i:
                        fs[i] += _pair_force(self.kind, self.P[i], ent_p[i], ent_v[i], ent_r[i], ent_p[j], ent_v[j], ent_r[j])
        ydot = np.empty((n, 6 if self.headed else 4))
        for i in range(n):
            P = self.P[i]
            pos, vel, radius, mass, vd = r[i, 0:2], r[i, 3:5], r[i, 8], r[i, 9], r[i, 12]
            diff = self.goals[i][0] - pos
            dist = np.linalg.norm(diff)
            if dist > radius:
                r[i, 14:16] = mass * (diff / dist * vd - vel) / P[0]
            fd = r[i, 14:16]
            fo = np.zeros(2)
            for o in obstacles[i]:
                d = pos - o
                dn = np.linalg.norm(d)
                n_iw = d / dn
                t_iw = np.array([-n_iw[1], n_iw[0]])
                dv = -np.dot(vel, t_iw)
                rd = rs[i] - dn
                if self.kind == 1:
                    fo += (P[2] * np.exp(rd / P[4]) + P[10] * max(0, rd)) * n_iw + (-P[6] * np.exp(rd / P[8]) - P[11] * max(0, rd)) * dv * t_iw
                else:
                    fo += (P[2] * np.exp(rd / P[4]) + P[10] * max(0, rd)) * n_iw - P[11] * max(0, rd) * dv * t_iw
            if obstacles[i] and self.kind != 1:
                fo /= len(obstacles[i])
            if not self.headed:
                gf = fd + fo + fs[i]
                ydot[i] = [vel[0], vel[1], gf[0] / mass, gf[1] / mass]
            else:
                inertia = 0.5 * mass * radius * radius
                tot = fd + fo + fs[i] if self.torque_new else fd
                tn = np.linalg.norm(tot)
                k_theta = inertia * P[19] * tn
                k_omega = inertia * (1 + P[18]) * np.sqrt(P[19] * tn / P[18])
                torque = -k_theta * _bound_angle(r[i, 2] - np.arctan2(tot[1], tot[0])) - k_omega * r[i, 7]
                c, s = np.cos(r[i, 2]), np.sin(r[i, 2])
                R = np.array([[c, -s], [s, c]])
                g0 = np.dot(fd + fo + fs[i], R[:, 0])
                g1 = P[16] * np.dot(fo + fs[i], R[:, 1]) - P[17] * r[i, 6]
                bv = r[i, 5:7]
                ydot[i] = [np.dot(R[0, :], bv), np.dot(R[1, :], bv), r[i, 7], g0 / mass, g1 / mass, torque / inertia]
        return ydot.ravel()

    def rhs(self, t, y):
        self.nfev += 1
        self._set_state(y)
        return self._forces()

    def update_humans(self, t, dt):
        from scipy.integrate import solve_ivp

        r = self.rows
        self.nfev = 0
        y0 = (r[:, [0, 1, 2, 5, 6, 7]] if self.headed else r[:, [0, 1, 3, 4]]).ravel().copy()
        sol = solve_ivp(self.rhs, (t, t + dt), y0, method="RK45")
        self._set_state(sol.y[:, -1])
        if self.headed:
            for i in range(len(r)):
                c, s = np.cos(r[i, 2]), np.sin(r[i, 2])
                r[i, 3:5] = np.array([[c, -s], [s, c]]) @ r[i, 5:7]
        for i in range(len(r)):
            r[i, 10:12] = self.goals[i][0]
        return self.nfev

    def complete_simulation(self, t, dt, final_time):
        """MotionModelManager.complete_rk45_simulation(t, dt, final_time) (motion_model_manager.py:461-498): ONE solve over
        (t, t + final_time) with t_eval = arange(t, final_time, dt) -> human_states [len(t_eval), n, 6 | 4] (the raw solution
        components).  The rows are left as the LAST right-hand-side evaluation leaves them (the reference does not set them
        from the solution afterwards)."""
        from scipy.integrate import solve_ivp

        r = self.rows
        self.nfev = 0
        y0 = (r[:, [0, 1, 2, 5, 6, 7]] if self.headed else r[:, [0, 1, 3, 4]]).ravel().copy()
        times = np.arange(t, final_time, dt, dtype=np.float64)
        sol = solve_ivp(self.rhs, (t, t + final_time), y0, method="RK45", t_eval=times)
        ns = 6 if self.headed else 4
        out = sol.y.T.reshape(len(times), len(r), ns).copy()
        for i in range(len(r)):
            r[i, 10:12] = self.goals[i][0]
        return out

    def respawn(self, bound_x, bound_y):
        """The parallel-traffic respawn rule on the agent objects (motion_model_manager.py:405-422, `self.parallel` False): the humans
        within 3 m of their goal are moved behind everybody else in index order; position and goal list only."""
        r = self.rows
        for i in range(len(r)):
            g0 = self.goals[i][0]
            if np.linalg.norm(r[i, 0:2] - g0) < 3:
                xs = [r[j, 0] for j in range(len(r))]
                rs = [r[j, 8] + r[j, 13] for j in range(len(r))]
                if self.robot is not None:
                    xs.append(self.robot[0]); rs.append(self.robot[4] + self.robot[5])
                r[i, 0] = max(max(xs) + max(rs) * 2, bound_x)
                r[i, 1] = min(r[i, 1], bound_y) if r[i, 1] >= 0 else max(r[i, 1], -bound_y)
                self.goals[i] = [np.array([g0[0], r[i, 1]])]
                r[i, 10:12] = self.goals[i][0]


class Rk45Robot:
    """update_robot(t, dt) of a robot whose SFM / HSFM model runs under RK45 (motion_model_manager.py:631-640): scipy's solve_ivp
    around f_rk45_robot_* (:661-687) = the trial state written into the robot (:88-103), compute_robot_forces (:591-613: the force
    part of robot_model_substep above, humans standing), ydot.  row as in robot_model_substep."""

    def __init__(self, row, P, model, hum_pos, hum_vel, hum_radius, hum_safety, walls):
        self.row = np.array(row, dtype=np.float64)
        self.P, self.model = np.asarray(P, dtype=np.float64), model
        self.h = (np.asarray(hum_pos, np.float64), np.asarray(hum_vel, np.float64), np.asarray(hum_radius, np.float64), np.asarray(hum_safety, np.float64))
        self.walls = walls
        self.headed = model.startswith("hsfm")
        self.nfev = 0

    def _set_state(self, y):
        r = self.row
        r[0:2] = y[0:2]
        if self.headed:
            r[2] = _bound_angle(y[2])
            bv = np.array(y[3:5])
            sp = np.linalg.norm(bv)
            r[5:7] = bv / sp * r[12] if sp > r[12] else bv
            r[7] = y[5]
        else:
            v = np.array(y[2:4])
            sp = np.linalg.norm(v)
            r[3:5] = v / sp * r[12] if sp > r[12] else v

    def rhs(self, t, y):
        self.nfev += 1
        self._set_state(y)
        r = self.row
        vel, R, fd, fo, fs = _robot_forces(r, self.P, self.model, *self.h, self.walls)
        r[3:5] = vel
        r[14:16] = fd
        P, mass, radius = self.P, r[9], r[8]
        if not self.headed:
            gf = fd + fo + fs
            return np.array([vel[0], vel[1], gf[0] / mass, gf[1] / mass])
        inertia = 0.5 * mass * radius * radius
        tot = fd if self.model in ("hsfm_farina", "hsfm_guo", "hsfm_moussaid") else fd + fo + fs
        tn = np.linalg.norm(tot)
        k_theta = inertia * P[19] * tn
        k_omega = inertia * (1 + P[18]) * np.sqrt(P[19] * tn / P[18])
        torque = -k_theta * _bound_angle(r[2] - np.arctan2(tot[1], tot[0])) - k_omega * r[7]
        g0 = np.dot(fd + fo + fs, R[:, 0])
        g1 = P[16] * np.dot(fo + fs, R[:, 1]) - P[17] * r[6]
        bv = r[5:7]
        return np.array([np.dot(R[0, :], bv), np.dot(R[1, :], bv), r[7], g0 / mass, g1 / mass, torque / inertia])

    def update(self, t, dt):
        from scipy.integrate import solve_ivp

        r = self.row
        self.nfev = 0
        y0 = r[[0, 1, 2, 5, 6, 7]].copy() if self.headed else r[[0, 1, 3, 4]].copy()
        sol = solve_ivp(self.rhs, (t, t + dt), y0, method="RK45")
        self._set_state(sol.y[:, -1])
        if self.headed:
            c, s = np.cos(r[2]), np.sin(r[2])
            r[3:5] = np.array([[c, -s], [s, c]]) @ r[5:7]
        return self.nfev
