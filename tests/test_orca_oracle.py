"""CPU suite for the ORCA restatement (oracle/orca_oracle.c).  PARITY UNPINNED: the reference calls the
third-party rvo2 library, which is absent; these tests pin the restatement to analytic known answers and to
a brute-force f64 solver of the same half-plane programme."""
import itertools
import math

import numpy as np

from oracle import crowd_oracle as orc


def test_single_agent_takes_pref_velocity_clipped_to_max_speed():
    v = orc.orca_new_velocities([[0, 0]], [[0, 0]], [[0.3, 0.4]], [0.31], [1.0])
    np.testing.assert_allclose(v, [[0.3, 0.4]], atol=1e-7)
    v = orc.orca_new_velocities([[0, 0]], [[0, 0]], [[3.0, 4.0]], [0.31], [1.0])
    np.testing.assert_allclose(v, [[0.6, 0.8]], atol=1e-6)


def test_far_agents_do_not_interact():
    pos = [[0, 0], [20, 0]]  # beyond neighborDist = 10
    v = orc.orca_new_velocities(pos, [[1, 0], [-1, 0]], [[1, 0], [-1, 0]], [0.31, 0.31], [1, 1])
    np.testing.assert_allclose(v, [[1, 0], [-1, 0]], atol=1e-7)


def test_head_on_pair_is_mirror_symmetric_and_each_takes_half():
    """Two agents head-on: the two solutions mirror each other (reciprocity: each takes u/2), each
    satisfies its own half-plane with equality or better, and the relative velocity leaves the VO."""
    pos = np.array([[-2.0, 0.0], [2.0, 0.0]])
    vel = np.array([[1.0, 0.0], [-1.0, 0.0]])
    v, lines, nl = orc.orca_new_velocities(pos, vel, vel, [0.31, 0.31], [1, 1], time_step=0.25, return_lines=True)
    assert nl.tolist() == [1, 1]
    np.testing.assert_allclose(v[0], -v[1], atol=1e-6)           # point symmetry of the scene
    for a in range(2):
        px, py, dx, dy = lines[a, 0]
        assert dx * (py - v[a, 1]) - dy * (px - v[a, 0]) <= 1e-6  # det(dir, point - v) <= 0: feasible
    # the pair no longer collides within the time horizon tau = 5: |rel_pos + t * rel_vel| >= R for t in [0, 5]
    rp, rv, R = pos[1] - pos[0], v[1] - v[0], 0.62
    ts = np.linspace(0, 5, 2001)
    d = np.linalg.norm(rp[None] + ts[:, None] * rv[None], axis=1)
    assert d.min() >= R - 1e-4


def test_colliding_pair_uses_the_time_step_cutoff():
    pos = np.array([[0.0, 0.0], [0.4, 0.0]])  # R = 0.62 > 0.4: already colliding
    vel = np.zeros((2, 2))
    dt = 0.25
    v = orc.orca_new_velocities(pos, vel, vel, [0.31, 0.31], [1, 1], time_step=dt)
    # w = rv - rp/dt = (-1.6, 0); u = (R/dt - |w|) * w/|w| = (0.88)(-1,0); each takes half: v0 = -0.44 x
    np.testing.assert_allclose(v[0], [-0.44, 0.0], atol=1e-6)
    np.testing.assert_allclose(v[1], [0.44, 0.0], atol=1e-6)


def _brute_force(lines, radius, pref):
    """Closest point to pref inside the disc and all half-planes det(dir, point - v) <= 0 (f64).
    Candidates: pref (clipped), projections on lines, line-line and line-circle intersections."""
    pts = []
    p = np.array(pref, float)
    if np.linalg.norm(p) > radius:
        p = p / np.linalg.norm(p) * radius
    pts.append(p)
    L = [(np.array(l[:2], float), np.array(l[2:], float)) for l in lines]
    for pt, d in L:
        t = np.dot(d, np.array(pref) - pt)
        pts.append(pt + t * d)
        b = np.dot(pt, d); c = np.dot(pt, pt) - radius ** 2
        disc = b * b - c
        if disc >= 0:
            for sgn in (-1, 1):
                pts.append(pt + (-b + sgn * math.sqrt(disc)) * d)
    for (p1, d1), (p2, d2) in itertools.combinations(L, 2):
        den = d1[0] * d2[1] - d1[1] * d2[0]
        if abs(den) > 1e-9:
            t = ((p2[0] - p1[0]) * d2[1] - (p2[1] - p1[1]) * d2[0]) / den
            pts.append(p1 + t * d1)
    best, bd = None, 1e18
    for q in pts:
        if np.linalg.norm(q) > radius + 1e-6:
            continue
        if all(d[0] * (pt[1] - q[1]) - d[1] * (pt[0] - q[0]) <= 1e-6 for pt, d in L):
            dist = np.linalg.norm(q - np.array(pref))
            if dist < bd:
                best, bd = q, dist
    return best


def test_lp_result_matches_brute_force_on_random_scenes():
    rng = np.random.default_rng(7)
    checked = feasible = 0
    for trial in range(300):
        na = int(rng.integers(2, 12))
        pos = rng.uniform(-3, 3, (na, 2))
        while True:  # no initial overlaps: keeps most programmes feasible
            d = np.linalg.norm(pos[:, None] - pos[None], axis=-1) + np.eye(na) * 9
            if d.min() > 0.7:
                break
            pos = rng.uniform(-3, 3, (na, 2))
        vel = rng.normal(0, 0.5, (na, 2))
        pref = rng.normal(0, 0.7, (na, 2))
        v, lines, nl = orc.orca_new_velocities(pos, vel, pref, np.full(na, 0.31), np.ones(na), time_step=0.0125,
                                               return_lines=True)
        assert np.all(np.linalg.norm(v, axis=1) <= 1.0 + 1e-5)
        for a in range(na):
            ref = _brute_force(lines[a, :nl[a]].astype(float), 1.0, pref[a])
            checked += 1
            if ref is not None:  # feasible programme: LP2's answer is the unique closest point
                feasible += 1
                assert np.linalg.norm(v[a] - ref) < 2e-4, (trial, a, v[a], ref)
    assert feasible > 0.9 * checked


def test_infeasible_programme_minimises_the_maximum_violation():
    """Agent squeezed between two head-on approachers: LP3 returns a velocity inside the disc whose worst
    half-plane violation is not larger than that of a fine grid search."""
    pos = np.array([[0.0, 0.0], [-0.7, 0.0], [0.7, 0.0], [0.0, 0.7], [0.0, -0.7]])
    vel = np.array([[0.0, 0.0], [1.0, 0.0], [-1.0, 0.0], [0.0, -1.0], [0.0, 1.0]])
    v, lines, nl = orc.orca_new_velocities(pos, vel, vel, np.full(5, 0.31), np.ones(5), time_step=0.25, return_lines=True)
    L = lines[0, :nl[0]].astype(float)

    def worst(q):
        return max(d[2] * (d[1] - q[1]) - d[3] * (d[0] - q[0]) for d in L)

    g = np.linspace(-1, 1, 201)
    grid_best = min(worst((x, y)) for x in g for y in g if x * x + y * y <= 1.0)
    assert np.linalg.norm(v[0]) <= 1.0 + 1e-5
    assert worst(v[0]) <= grid_best + 2e-2


def test_step_block_goal_rotation_pref_velocity_and_robot_lag():
    # one human walking to a goal 0.5 m away with strict-< rotation; pref vel is the raw difference within vd
    S = np.zeros((1, 13), np.float32)
    S[0, 0:2] = [0.0, 0.0]; S[0, 8] = 0.3; S[0, 12] = 1.0
    goals = np.array([[[0.5, 0.0], [-3.0, 0.0]]], np.float32)
    S[0, 10:12] = goals[0, 0]; S[0, 5:7] = [0.5, 0.0]
    S2, g2, _ = orc.orca_step_block(S, goals, [0.01], 0.25, 1)
    np.testing.assert_allclose(S2[0, 3:5], [0.5, 0.0], atol=1e-7)      # pref vel 0.5 < max speed
    np.testing.assert_allclose(S2[0, 0:2], [0.125, 0.0], atol=1e-7)
    np.testing.assert_allclose(S2[0, 5:7], [0.375, 0.0], atol=1e-6)    # |g - p| <= vd: not normalised (:131)
    S3, g3, _ = orc.orca_step_block(S2, g2, [0.01], 0.25, 2)           # reaches |g - p| < r -> rotate
    assert np.allclose(g3[0, 0], [-3.0, 0.0]) and np.allclose(g3[0, 1], [0.5, 0.0])
    assert abs(np.linalg.norm(S3[0, 5:7]) - 1.0) < 1e-6                # far goal: unit preferred velocity
