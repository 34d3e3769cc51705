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
        # float32: with two nearly anti-parallel half-planes linearProgram3's projected line is ill-conditioned and the point
        # it returns on the speed circle is off by ~6e-5 (one such scene in 300, either side of the circle)
        assert np.all(np.linalg.norm(v, axis=1) <= 1.0 + 2e-4)
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


# ------------------------------------------------------------------ static obstacles
def test_obstacle_preprocessing_known_answers():
    sq = orc.process_obstacle([[0, 0], [2, 0], [2, 2], [0, 2]])                 # counter-clockwise square
    assert sq[:, 4].tolist() == [1, 1, 1, 1]
    np.testing.assert_allclose(sq[:, 2:4], [[1, 0], [0, 1], [-1, 0], [0, -1]], atol=1e-7)
    assert sq[:, 5].tolist() == [1, 2, 3, 0] and sq[:, 6].tolist() == [3, 0, 1, 2]
    ell = orc.process_obstacle([[0, 0], [2, 0], [2, 1], [1, 1], [1, 2], [0, 2]])  # L shape: one reflex vertex (1, 1)
    assert ell[:, 4].tolist() == [1, 1, 1, 0, 1, 1]
    two = orc.process_obstacles([[[0, 0], [1, 0], [0, 1]], [[5, 5], [6, 5], [6, 6], [5, 6]]])
    assert two[:, 5].tolist() == [1, 2, 0, 4, 5, 6, 3] and two[3:, 6].tolist() == [6, 3, 4, 5]
    line = orc.process_obstacle([[3, 0], [-3, 0]])                                # two vertices: always convex
    assert line[:, 4].tolist() == [1, 1]


def test_agent_walking_into_a_wall_is_slowed_to_the_cutoff_speed():
    """Wall along the x axis, agent above it heading straight down: the cut-off line of the wall's velocity obstacle
    gives v_y >= -(d - r) / timeHorizonObst, so the new velocity is exactly that (its preferred speed is larger)."""
    wall = orc.process_obstacle([[6, 0], [-6, 0]])     # the edge (6,0) -> (-6,0) has the agent on its right
    d, r, tau = 2.0, 0.31, 5.0
    v, lines, nl, no = orc.orca_new_velocities_obst([[0.3, d]], [[0, -1]], [[0, -1]], [r], [1.0], wall, time_horizon_obst=tau)
    assert no[0] == 1 and nl[0] == 1
    np.testing.assert_allclose(lines[0, 0], [0.3 * 0 + (6 - 0.3) / tau, (r - d) / tau, 1.0, 0.0], atol=1e-6)
    np.testing.assert_allclose(v[0], [0.0, -(d - r) / tau], atol=1e-6)
    # far from the wall (beyond timeHorizonObst * maxSpeed + radius) it is not even a neighbour
    v, _, nl, no = orc.orca_new_velocities_obst([[0.0, 6.0]], [[0, -1]], [[0, -1]], [r], [1.0], wall, time_horizon_obst=tau)
    assert no[0] == 0 and np.allclose(v[0], [0, -1])
    # on the other side of the directed edge the wall does not exist for the agent (RVO2: polygons are one-sided)
    v, _, nl, no = orc.orca_new_velocities_obst([[0.0, -2.0]], [[0, 1]], [[0, 1]], [r], [1.0], wall[:1], time_horizon_obst=tau)
    assert np.allclose(v[0], [0, 1])
    # touching the wall: the line through the origin along -unitDir, only sliding (or leaving) is allowed
    v, lines, nl, no = orc.orca_new_velocities_obst([[0.0, 0.25]], [[0.5, -0.5]], [[0.5, -0.5]], [r], [1.0], wall)
    np.testing.assert_allclose(lines[0, 0], [0, 0, 1, 0], atol=1e-7)
    np.testing.assert_allclose(v[0], [0.5, 0.0], atol=1e-6)


def _polys(rng):
    out = []
    for k in range(int(rng.integers(1, 4))):
        c = rng.uniform(-3, 3, 2)
        nv = int(rng.integers(3, 6))
        ang = np.sort(rng.uniform(0, 2 * np.pi, nv))
        if np.min(np.diff(np.append(ang, ang[0] + 2 * np.pi))) < 0.5:
            ang = np.linspace(0, 2 * np.pi, nv, endpoint=False) + rng.uniform(0, 1)
        rad = rng.uniform(0.4, 1.0, nv)
        out.append([[float(c[0] + rad[i] * np.cos(ang[i])), float(c[1] + rad[i] * np.sin(ang[i]))] for i in range(nv)])  # CCW
    return out


def _inside_or_near(p, poly, margin):
    poly = np.array(poly)
    n = len(poly)
    inside = all((poly[(i + 1) % n][0] - poly[i][0]) * (p[1] - poly[i][1]) - (poly[(i + 1) % n][1] - poly[i][1]) * (p[0] - poly[i][0]) >= 0
                 for i in range(n))
    if inside:
        return True
    for i in range(n):
        a, b = poly[i], poly[(i + 1) % n]
        t = np.clip(np.dot(p - a, b - a) / np.dot(b - a, b - a), 0, 1)
        if np.linalg.norm(p - (a + t * (b - a))) < margin:
            return True
    return False


def test_obstacle_lines_are_hard_constraints_and_the_lp_matches_brute_force():
    """Random agents among random convex-ish polygons: every obstacle line is satisfied by the returned velocity (they are
    hard constraints of linearProgram3 too), and where the whole programme is feasible the velocity is the unique closest
    point to the preferred velocity (brute-force f64 vertex enumeration over obstacle + agent lines)."""
    rng = np.random.default_rng(11)
    with_obst = feasible = checked = 0
    for trial in range(250):
        polys = _polys(rng)
        verts = orc.process_obstacles(polys)
        na = int(rng.integers(1, 9))
        pos = []
        while len(pos) < na:
            p = rng.uniform(-4, 4, 2)
            if all(np.linalg.norm(p - q) > 0.7 for q in pos) and not any(_inside_or_near(p, poly, 0.36) for poly in polys):
                pos.append(p)
        pos = np.array(pos)
        vel = rng.normal(0, 0.5, (na, 2))
        pref = rng.normal(0, 0.8, (na, 2))
        v, lines, nl, no = orc.orca_new_velocities_obst(pos, vel, pref, np.full(na, 0.31), np.ones(na), verts, time_step=0.0125)
        assert np.all(np.linalg.norm(v, axis=1) <= 1.0 + 2e-4)
        for a in range(na):
            L = lines[a, :nl[a]].astype(float)
            with_obst += no[a] > 0
            ref = _brute_force(L, 1.0, pref[a])
            checked += 1
            if ref is not None:
                feasible += 1
                assert np.linalg.norm(v[a] - ref) < 3e-4, (trial, a, v[a], ref)
            obst_feasible = _brute_force(L[:no[a]], 1.0, pref[a]) is not None
            if obst_feasible:   # the obstacle half-planes alone admit a velocity: none of them may be violated
                for px, py, dx, dy in L[:no[a]]:
                    assert dx * (py - v[a, 1]) - dy * (px - v[a, 0]) <= 2e-4, (trial, a)
    assert with_obst > 200 and feasible > 0.8 * checked


def test_agent_does_not_cross_a_wall_over_many_steps():
    """An agent whose goal lies straight behind a box walks up to it and stops / slides along it (ORCA is local: it does
    not plan around): it never comes closer to the polygon than its radius (minus float slack), in 120 steps."""
    verts = orc.process_obstacles([[[-1, -0.5], [1, -0.5], [1, 0.5], [-1, 0.5]]])
    for x0 in (0.2, 0.9, 1.25):
        S = np.zeros((1, 1, 13), np.float32)
        S[0, 0, 0:2] = [x0, 3.0]; S[0, 0, 8] = 0.3; S[0, 0, 12] = 1.0
        goals = np.array([[[[x0 + 0.1, -3.0]]]], np.float32)
        S[0, 0, 10:12] = goals[0, 0, 0]; S[0, 0, 5:7] = [0.0, -1.0]
        closest = 1e9
        for k in range(120):
            S, goals, _ = orc.orca_step_block(S, goals, [[0.01]], 0.1, 1, verts=verts)
            p = S[0, 0, 0:2].astype(float)
            d = math.hypot(max(abs(p[0]) - 1.0, 0.0), max(abs(p[1]) - 0.5, 0.0))    # distance to the box
            assert d > 0.31 - 5e-3, (x0, k, p, d)
            closest = min(closest, d)
        if x0 < 0.5:   # head-on: v_n = -(d - r) / tau per step -> the gap decays geometrically, d_k - r = (d_0 - r)(1 - dt/tau)^k
            np.testing.assert_allclose(closest, 0.31 + (2.5 - 0.31) * (1 - 0.1 / 5.0) ** 120, rtol=5e-3)
        if x0 > 1.2:   # the path beside the box is free: it gets past
            assert S[0, 0, 1] < -1.0


def test_per_agent_rvo2_parameters_in_the_restatement():
    """RVO2's per-agent neighborDist / maxNeighbors / timeHorizon (RVOSimulator::addAgent's arguments) in the C restatement: the uniform
    array equals the scalar call bit for bit; maxNeighbors = 0 or a neighborDist shorter than the gap makes an agent blind (it keeps its
    preferred velocity while its partner, who still sees it, takes the whole avoidance); a longer timeHorizon reacts earlier."""
    from oracle import crowd_oracle as orc

    def head_on(ap):
        S = np.zeros((1, 2, 13), np.float32)
        S[0, 0, 0:2] = (-2.0, 0.02); S[0, 1, 0:2] = (2.0, -0.02)
        S[0, 0, 3:5] = S[0, 0, 5:7] = (1.0, 0.0); S[0, 1, 3:5] = S[0, 1, 5:7] = (-1.0, 0.0)
        S[0, :, 8] = 0.3; S[0, :, 12] = 1.0
        goals = np.full((1, 2, 1, 2), 1.0e9, np.float32)
        out, _, _ = orc.orca_step_block(S, goals, np.zeros((1, 2), np.float32), 0.25, 1, agent_params=ap)
        return out[0, :, 3:5]

    uniform = np.tile(np.array([10.0, 10, 5.0, 5.0], np.float32), (1, 2, 1))
    np.testing.assert_array_equal(head_on(uniform), head_on(None))
    both = head_on(None)
    assert abs(both[0, 1]) > 1e-3 and abs(both[1, 1]) > 1e-3                 # reciprocal: both give way
    for blind in (np.array([10.0, 0, 5.0, 5.0], np.float32), np.array([3.0, 10, 5.0, 5.0], np.float32)):   # no neighbours / too short a range (gap 4 m)
        ap = uniform.copy(); ap[0, 0] = blind
        v = head_on(ap)
        np.testing.assert_array_equal(v[0], np.array([1.0, 0.0], np.float32))  # agent 0 sees nobody: preferred velocity
        np.testing.assert_array_equal(v[1], both[1])                           # agent 1 still takes its half
    short, long_ = uniform.copy(), uniform.copy()
    short[0, :, 2] = 0.5; long_[0, :, 2] = 8.0                                 # time to collision: 1.7 s
    assert np.max(np.abs(head_on(short)[:, 1])) < 1e-6 < np.min(np.abs(head_on(long_)[:, 1]))
