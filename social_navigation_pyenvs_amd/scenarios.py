"""Vectorised synthetic scenario generators for thousands of worlds (throughput runs).

These follow the *rules* of the reference generators (social_nav_sim.py:200-431: circular crossing
with position noise and a 0.2 m discomfort margin to other humans and their goals; parallel traffic
in a length x height box with a 0.1 m margin and a single goal at x = -L/2-3, respawn bounds
(L/2, H/2)) but draw counter-based uniforms keyed by (seed, GLOBAL world id), so a world is the same whatever
batch / shard / GPU count it is generated in, and they do not reproduce the legacy ``np.random`` stream -- parity of the generators themselves is covered by the exact host
restatement in social_gym/social_nav_sim.py, not here ("generator need not match
the reference RNG").  Rejection loops are bounded (the reference's are not).
"""
from __future__ import annotations

import math

import numpy as np

SFMS = ["sfm_helbing", "sfm_guo", "sfm_moussaid", "hsfm_farina", "hsfm_guo", "hsfm_moussaid",
        "hsfm_new", "hsfm_new_guo", "hsfm_new_moussaid"]

# agent.py:94-243 default model constants, as the 20-vector of agent.py:268-388
_BASE = dict(relax_t=0.5, Ai=2000.0, Aw=2000.0, Bi=0.08, Bw=0.08, Ci=120.0, Cw=120.0, Di=0.6, Dw=0.6, Ei=360.0,
             k1=120000.0, k2=240000.0, lam=2.0, gam=0.35, ns=2.0, ns1=3.0, ko=1.0, kd=500.0, alpha=3.0,
             klam=0.1)
_ORDER = ["relax_t", "Ai", "Aw", "Bi", "Bw", "Ci", "Cw", "Di", "Dw", "Ei", "k1", "k2", "lam", "gam", "ns", "ns1",
          "ko", "kd", "alpha", "klam"]
_USED = {
    "helbing": ["relax_t", "Ai", "Aw", "Bi", "Bw", "k1", "k2"],
    "guo": ["relax_t", "Ai", "Aw", "Bi", "Bw", "Ci", "Cw", "Di", "Dw", "k1", "k2"],
    "moussaid": ["relax_t", "Ei", "lam", "gam", "ns", "ns1", "Aw", "Bw", "k1", "k2"],
}
_HEADED = ["ko", "kd", "alpha", "klam"]


def default_params(model: str | int) -> np.ndarray:
    """The [20] parameter row Agent.get_parameters(model) returns for default constants."""
    if isinstance(model, int):
        model = SFMS[model]
    if model not in SFMS:
        raise ValueError(f"unknown model {model}")
    kind = "guo" if model.endswith("guo") else ("moussaid" if model.endswith("moussaid") else "helbing")
    used = list(_USED[kind]) + (_HEADED if model.startswith("hsfm") else [])
    return np.array([_BASE[k] if k in used else 0.0 for k in _ORDER], dtype=np.float64)


_M64 = (1 << 64) - 1


def _u01(seed, wid, i, t, k):
    """Counter-based uniforms in [0, 1): a splitmix64 hash of (seed, GLOBAL world id, human, attempt, draw).  A world's
    scenario is a pure function of (seed, its global id): it does not depend on which other worlds share the batch, on
    the shard boundaries or on the number of GPUs (sharding.py) -- unlike one generator stream consumed by a vectorised
    rejection loop."""
    x = (np.asarray(wid, dtype=np.uint64) * np.uint64(0xBF58476D1CE4E5B9)
         + np.uint64((int(seed) * 0x9E3779B97F4A7C15 + int(i) * 0x94D049BB133111EB + int(t) * 0xD6E8FEB86659FD93
                      + int(k) * 0xA0761D6478BD642F) & _M64))
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _place(wid, n, sampler, ok, max_tries=400):
    """Sequential rejection sampling of n points per world, vectorised over worlds; `wid` = global world ids."""
    W = wid.size
    pts = np.zeros((W, n, 2))
    extra = [None] * n
    for i in range(n):
        todo = np.arange(W)
        tries = 0
        while todo.size:
            cand, ex = sampler(todo, i, tries)
            good = ok(cand, pts[todo, :i], todo, i)
            tries += 1
            if tries >= max_tries:
                good[:] = True  # bounded: accept (the reference would loop forever)
            sel = todo[good]
            pts[sel, i] = cand[good]
            if ex is not None:
                if extra[i] is None:
                    extra[i] = np.zeros(W)
                extra[i][sel] = ex[good]
            todo = todo[~good]
    return pts, extra


def circular_crossing(W, n, radius=7.0, seed0=1000, r=0.3, vd=1.0, first_world=0):
    """[W,n] circular crossing: pos = R(cos,sin) + U(-.5,.5)^2 * vd, goal = -pos, goals [-pos, pos].
    Worlds are numbered first_world .. first_world + W - 1; world g is the same whatever batch it is generated in."""
    wid = np.arange(first_world, first_world + W, dtype=np.uint64)
    with np.errstate(over="ignore"):
        def sampler(todo, i, t):
            ang = _u01(seed0, wid[todo], i, t, 0) * 2 * math.pi
            noise = (np.stack([_u01(seed0, wid[todo], i, t, 1), _u01(seed0, wid[todo], i, t, 2)], -1) - 0.5) * vd
            p = radius * np.stack([np.cos(ang), np.sin(ang)], -1) + noise
            return p, ang

        def ok(cand, placed, todo, i):
            if placed.shape[1] == 0:
                return np.ones(cand.shape[0], bool)
            d1 = np.linalg.norm(cand[:, None] - placed, axis=-1)
            d2 = np.linalg.norm(cand[:, None] + placed, axis=-1)  # other humans' goals (-pos)
            return (d1.min(1) >= 2 * r + 0.2) & (d2.min(1) >= 2 * r + 0.2)

        pos, ang = _place(wid, n, sampler, ok)
    ang = np.stack(ang, 1)
    yaw = np.mod(math.pi + ang + math.pi, 2 * math.pi) - math.pi
    goals = np.stack([-pos, pos], axis=2)  # [W,n,2,2]
    return pos, yaw, goals


def parallel_traffic(W, n, length=14.0, height=3.0, seed0=2000, r=0.3, first_world=0):
    """[W,n] parallel traffic: x ~ U(-L/2+r, L/2-r), y ~ U(-H/2, H/2), gap >= 0.1; goal (-L/2-3, y)."""
    if n * math.pi * r * r > length * height * 0.4:
        raise ValueError("Number of humans specified is too big for desided traffic height and length")
    wid = np.arange(first_world, first_world + W, dtype=np.uint64)
    with np.errstate(over="ignore"):
        def sampler(todo, i, t):
            a, b = -length / 2 + r, length / 2 - r
            p = np.stack([(b - a) * _u01(seed0, wid[todo], i, t, 0) + a, (_u01(seed0, wid[todo], i, t, 1) - 0.5) * height], -1)
            return p, None

        def ok(cand, placed, todo, i):
            if placed.shape[1] == 0:
                return np.ones(cand.shape[0], bool)
            d = np.linalg.norm(cand[:, None] - placed, axis=-1)
            return d.min(1) - 2 * r - 0.1 >= 0

        pos, _ = _place(wid, n, sampler, ok)
    yaw = np.full((W, n), -math.pi)
    goals = np.stack([np.full((W, n), -length / 2 - 3.0), pos[..., 1]], -1)[:, :, None, :]  # [W,n,1,2]
    return pos, yaw, goals


def make_states(pos, yaw, goals, r=0.3, m=75.0, vd=1.0):
    """Rows S[W,n,13] at rest (agent.py:256-258 layout) from positions / yaws / first goals."""
    W, n = pos.shape[:2]
    S = np.zeros((W, n, 13))
    S[..., 0:2] = pos
    S[..., 2] = yaw
    S[..., 8] = r
    S[..., 9] = m
    S[..., 10:12] = goals[:, :, 0]
    S[..., 12] = vd
    return S


def hybrid_worlds(W, n, model="hsfm_farina", radius=7.0, length=14.0, height=3.0, seed0=1000, first_world=0):
    """BASELINE cfg3: worlds with an even GLOBAL id are circular crossings, odd ones parallel traffic (with respawn).
    Goals are padded to G = 2 slots (traffic worlds: NaN second slot).  Returns (S, goals, params, respawn_bounds); the
    per-world respawn switch is `(first_world + arange(W)) % 2`."""
    gid = first_world + np.arange(W)
    S = np.zeros((W, n, 13))
    goals = np.full((W, n, 2, 2), np.nan)
    even, odd = np.flatnonzero(gid % 2 == 0), np.flatnonzero(gid % 2 == 1)
    # every world is generated under its own global id (one call per parity class; ids are strided by 2)
    for sel, kind in ((even, "c"), (odd, "t")):
        if sel.size == 0:
            continue
        g0, cnt = int(gid[sel[0]]), sel.size
        if kind == "c":
            p, y, g = _strided(circular_crossing, g0, cnt, n, radius, seed0)
            S[sel] = make_states(p, y, g)
            goals[sel] = g
        else:
            p, y, g = _strided(parallel_traffic, g0, cnt, n, length, height, seed0 + 1)
            S[sel] = make_states(p, y, g)
            goals[sel, :, 0:1] = g
    params = np.tile(default_params(model), (n, 1))
    return S, goals, params, (length / 2, height / 2)


def _strided(fn, g0, cnt, n, *args):
    """fn for the worlds g0, g0 + 2, ...: generate the covering contiguous range and keep every second world (a world
    is a function of its global id only, so the skipped ones cost time, nothing else)."""
    p, y, g = fn(2 * cnt - 1, n, *args, first_world=g0)
    return p[::2], y[::2], g[::2]


def static_obstacle_worlds(W, n, model="hsfm_farina", radius=14.0, seed0=1000, first_world=0, n_static=3):
    """BASELINE cfg5: circular crossing of n humans on a circle of `radius` (>= 10: 50 humans do not fit R = 7) with BOTH
    kinds of static obstacles: the first `n_static` humans are immobile -- desired speed 0, radius 0.8,
    both goals = own position, standing on the inner circle radius - 3, as circular_crossing_with_static_obstacles builds
    them (social_nav_sim.py:381-387, 416-417) -- and three shared polygon walls (polygon_walls()).
    Returns (S, goals, params, walls)."""
    pos, yaw, g = circular_crossing(W, n, radius, seed0, first_world=first_world)
    S = make_states(pos, yaw, g)
    wid = np.arange(first_world, first_world + W, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(n_static):
            ang = (2.0 * math.pi / n_static) * (i + 0.5 * (_u01(seed0 + 7, wid, i, 0, 0) - 0.5))
            p = (radius - 3.0) * np.stack([np.cos(ang), np.sin(ang)], -1)
            S[:, i, 0:2] = p
            S[:, i, 10:12] = p
            g[:, i, 0] = p
            g[:, i, 1] = p
    S[:, :n_static, 12] = 0.0
    S[:, :n_static, 8] = 0.8
    return S, g, np.tile(default_params(model), (n, 1)), polygon_walls()


def polygon_walls():
    """Three convex polygons (3, 4, 5 vertices) as an [O=3, Smax=5, 2, 2] NaN-padded segment array."""
    polys = [
        [[-2.0, -1.0], [-1.0, -1.2], [-1.4, -0.2]],
        [[1.0, 1.0], [2.2, 1.0], [2.2, 1.8], [1.0, 1.8]],
        [[-0.6, 2.4], [0.4, 2.2], [0.9, 3.0], [0.1, 3.7], [-0.8, 3.2]],
    ]
    smax = max(len(p) for p in polys)
    arr = np.full((len(polys), smax, 2, 2), np.nan)
    for i, p in enumerate(polys):
        for j in range(len(p)):
            a, b = p[j], p[(j + 1) % len(p)]
            arr[i, j, 0], arr[i, j, 1] = min(a, b), max(a, b)
    return arr
