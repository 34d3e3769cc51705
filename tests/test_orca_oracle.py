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


# ---------------------------------------------------------------------------------------------------------------------------------
# An INDEPENDENT anchor for the half-plane construction of Agent::computeNewVelocity.  Nothing below shares
# code or formulae with oracle/orca_oracle.c: the velocity obstacle of a pair for the horizon tau is built as a point set -- the union
# over t in (0, tau] of the discs D(rp / t, R / t), whose boundary is the front arc of the cut-off disc D(rp / tau, R / tau) between
# the two tangent points seen from the origin, and the two tangent rays leaving those points away from the origin -- and `u` is found
# by projecting the relative velocity on the three boundary pieces in float64 and taking the nearest point (van den Berg et al. 2009,
# section 4: "u is the vector from v_A - v_B to the closest point on the boundary of the velocity obstacle", n the outward normal there).
# The restatement's line must then pass through v + u / 2 with left normal n.  For two overlapping agents RVO2 takes the cut-off disc
# of ONE time step instead (the pair must separate within the step).
# ---------------------------------------------------------------------------------------------------------------------------------
def _nearest_on_vo_boundary(rp, rv, R, tau):
    """(q, n): closest point to rv on the boundary of the truncated cone, outward normal there.  float64, geometric."""
    c = rp / tau
    rho = R / tau
    d = np.linalg.norm(c)
    axis = c / d                                         # from the origin towards the disc centre
    phi = math.acos(rho / d)                             # angle at the centre between the direction to the origin and to a tangent point
    cands = []
    # piece 1: the front arc -- points c + rho * e with the angle between e and -axis at most phi
    w = rv - c
    e = w / np.linalg.norm(w)
    ang = math.atan2(-(axis[0] * e[1] - axis[1] * e[0]), -(axis @ e))    # signed angle from -axis to e
    a_arc = min(max(ang, -phi), phi)                                      # clamp to the arc: its end points are the tangent points
    rot = lambda v, a: np.array([v[0] * math.cos(a) - v[1] * math.sin(a), v[0] * math.sin(a) + v[1] * math.cos(a)])
    e_arc = rot(-axis, a_arc)
    cands.append((c + rho * e_arc, e_arc))
    # pieces 2, 3: the tangent rays.  Tangent point T = c + rho * rot(-axis, +-phi); the ray leaves T along T / |T|
    for sgn in (1.0, -1.0):
        T = c + rho * rot(-axis, sgn * phi)
        t_hat = T / np.linalg.norm(T)
        s = max(0.0, (rv - T) @ t_hat)
        q = T + s * t_hat
        nrm = np.array([t_hat[1], -t_hat[0]])
        if nrm @ (T - c) < 0:                            # outward = away from the cone's inside, i.e. the side of the disc's own normal at T
            nrm = -nrm
        cands.append((q, nrm))
    return min(cands, key=lambda qn: np.linalg.norm(qn[0] - rv))


def test_half_plane_construction_against_an_independent_projection_on_the_velocity_obstacle():
    import ctypes as C

    rng = np.random.default_rng(20251005)
    N = 120_000
    f = np.float32
    kinds = rng.integers(0, 4, N)          # 0 far / random, 1 relative velocity near the cut-off arc (apex side), 2 deep along a leg, 3 overlapping
    pos = np.zeros((N, 2, 2), f); vel = np.zeros((N, 2, 2), f); rad = np.zeros((N, 2), f); taus = np.zeros(N)
    for i in range(N):
        R = rng.uniform(0.4, 1.2)
        ra = rng.uniform(0.15, R - 0.15); rad[i] = (ra, R - ra)
        tau = float(rng.choice([2.0, 5.0, 10.0])); taus[i] = tau
        ang = rng.uniform(0, 2 * math.pi)
        Rf = float(f(rad[i, 0]) + f(rad[i, 1]))
        dist = rng.uniform(0.2 * Rf, 0.97 * Rf) if kinds[i] == 3 else rng.uniform(1.03 * Rf, 8.0)
        rp = dist * np.array([math.cos(ang), math.sin(ang)])
        if kinds[i] == 1:
            rv = rp / tau + rng.normal(0, 1.5 * Rf / tau, 2)
        elif kinds[i] == 2:
            rv = rp / tau * rng.uniform(1.0, 6.0) + rng.normal(0, 0.5, 2)
        else:
            rv = rng.uniform(-2.5, 2.5, 2)
        p0 = rng.uniform(-3, 3, 2); v1 = rng.uniform(-1, 1, 2)
        pos[i, 0] = p0; pos[i, 1] = p0 + rp; vel[i, 1] = v1; vel[i, 0] = v1 + rv
    out = np.zeros((2, 2), f); lines = np.zeros((2, 10, 4), f); nl = np.zeros(2, np.int32)
    ones = np.ones(2, f)
    fn = orc.lib().orc_orca_new_velocities
    fn.restype = None
    fp = C.POINTER(C.c_float)
    worst_p = worst_n = 0.0
    STATS = []
    count = {0: 0, 1: 0, 2: 0, 3: 0}
    pieces = {"arc": 0, "leg": 0}
    dt = 0.0125
    for i in range(N):
        tau = taus[i]
        fn(C.c_int(2), pos[i].ctypes.data_as(fp), vel[i].ctypes.data_as(fp), vel[i].ctypes.data_as(fp), rad[i].ctypes.data_as(fp), ones.ctypes.data_as(fp),
           C.c_float(100.0), C.c_int(10), C.c_float(tau), C.c_float(dt), out.ctypes.data_as(fp), lines.ctypes.data_as(C.c_void_p), nl.ctypes.data_as(C.POINTER(C.c_int)))
        assert nl[0] == 1
        # the float32 inputs the restatement saw, exactly, in float64
        rp = pos[i, 1].astype(np.float64) - pos[i, 0].astype(np.float64)
        rv = vel[i, 0].astype(np.float64) - vel[i, 1].astype(np.float64)
        R = float(rad[i, 0]) + float(rad[i, 1])
        v = vel[i, 0].astype(np.float64)
        d = np.linalg.norm(rp)
        if d <= R * (1 + 1e-6) and d >= R * (1 - 1e-6):
            continue                                     # touching within float32 rounding of R: either construction is right
        if d > R:
            # skip what float32 cannot resolve: rv within 1e-4 of the disc centre (direction of w undefined), within 1e-4 rad of an arc end
            # (arc or leg: both give the same line there up to that angle) or of the cone's axis deep inside it (which leg)
            c = rp / tau
            if np.linalg.norm(rv - c) < 1e-3 * R / tau:
                continue
            q, n = _nearest_on_vo_boundary(rp, rv, R, tau)
            axis_side = abs(rp[0] * rv[1] - rp[1] * rv[0]) / d
            onleg = abs(n @ (q / max(np.linalg.norm(q), 1e-30))) < 1e-9 and np.linalg.norm(q - c) > R / tau * (1 + 1e-9)
            if onleg and axis_side < 1e-4:
                continue
            pieces["leg" if onleg else "arc"] += 1
        else:
            c = rp / dt
            w = rv - c
            n = w / np.linalg.norm(w)
            q = c + (R / dt) * n
        u = q - rv
        px, py, dx, dy = lines[0, 0].astype(np.float64)
        left_normal = np.array([-dy, dx])
        scale = max(1.0, np.linalg.norm(u), np.linalg.norm(rv), np.linalg.norm(v))
        ep = np.linalg.norm(np.array([px, py]) - (v + 0.5 * u)) / scale
        en = np.linalg.norm(left_normal - n)
        # near a tangent point the arc's and the leg's line differ by the angle between their normals: compare with the closer of the two
        # float32 conditioning of the construction itself: the arc's normal is w / |w| (w = rv - rp / tau carries ~1e-7 of absolute rounding), a leg's
        # direction comes from sqrt(d^2 - R^2) (relative rounding amplified by d^2 / (d^2 - R^2))
        if d > R:
            cond = d * d / (d * d - R * R) if onleg else 1.0 / max(np.linalg.norm(rv - rp / tau), 1e-30)
        else:
            cond = 1.0 / max(np.linalg.norm(rv - rp / dt) * dt, 1e-30)
        tol = 2e-6 * (1.0 + cond)
        worst_p = max(worst_p, ep / tol); worst_n = max(worst_n, en / tol)
        STATS.append((int(kinds[i]), ep, en, cond))
        assert ep < tol and en < tol, (i, int(kinds[i]), ep, en, tol, rp, rv, R, tau, lines[0, 0], q, n)
        count[int(kinds[i])] += 1
    assert sum(count.values()) > 0.98 * N and min(count.values()) > 0.2 * N and min(pieces.values()) > 0.1 * N, (count, pieces)
    print(f"half-plane anchor: {sum(count.values())} pairs {count} {pieces}, worst |point - (v + u/2)| / tolerance {worst_p:.2f}, worst |normal - n| / tolerance {worst_n:.2f}; "
          f"with cond < 10 ({sum(1 for s_ in STATS if s_[3] < 10)} pairs): worst point {max(s_[1] for s_ in STATS if s_[3] < 10):.2e}, normal {max(s_[2] for s_ in STATS if s_[3] < 10):.2e}")


def test_probe_instantiation_is_the_restatement_when_unperturbed_and_names_recorded_edge_cases():
    """oracle/orca_oracle_probe.c: (i) probe 0 (no noise) returns the restatement's velocity bit for bit on random crowds; (ii) on the agent-substeps
    RECORDED ON THE MI355X (tests/golden/orca_fast_edge_cases.npz: the world's input rows and the fast / fma build's answer where it was beyond
    1e-5 of the exact restatement and the double evaluation did not explain it -- all 29 the input-noise probes of round 5 left unexplained (10 fma + 19 fast), and a
    sample of every decision class) the restatement's own answer moves >= 1e-5 under <= 4-ulp operation noise, a decision is named, and some
    probe lands within 1e-5 of the GPU build's answer."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_io import load_cases
    import orca_fast_parity as ofp

    S, g, margin = ofp.crossing(6, 25, 7.0, 99)
    ref = S.copy(); rg = g.copy()
    for k in range(12):
        nxt, ng, _ = orc.orca_step_block(ref, rg, margin, 0.0125, 1)
        for w in range(6):
            for a in (0, 7, 24):
                v, tr = orc.orca_probe_agent(ref[w], margin[w], a, 0.0125, probes=1)
                assert np.array_equal(v[0], nxt[w, a, 3:5]) and len(tr[0][0]) > 0
        ref, rg = orc.orca_step_block(ref, rg, margin, 0.0125, 25)[:2]
    cases = load_cases("orca_fast_edge_cases")
    assert len(cases) >= 40 and sum(c["input_noise_class"] == "unexplained" for c in cases) == 29
    labels = {}
    for c in cases:
        v, _ = orc.orca_probe_agent(c["S"], c["margin"], c["agent"], 0.0125, probes=1)
        assert np.array_equal(v[0], c["exact_v"])                       # the recorded exact answer is this restatement's
        assert np.abs(c["build_v"].astype(np.float64) - c["exact_v"]).max() >= 1e-5 - 1e-12
        sensitive, label, reproduced = ofp.probe_classify(orc, c["S"], c["margin"], c["agent"], 0.0125, c["build_v"])
        assert sensitive and label not in ("not sensitive", "trace diverged"), (c["world"], c["substep"], c["agent"], label)
        assert reproduced or c["err"] < 2e-5, (c["world"], c["substep"], c["agent"], label, c["err"])
        labels[label] = labels.get(label, 0) + 1
    assert len(labels) >= 5, labels


def test_obstacle_kdtree_edge_splitting_two_restatements_agree_and_pieces_tile_the_edges():
    """RVO2's processObstacles() (KdTree::buildObstacleTreeRecursive) cuts every edge that straddles the line of a node's splitting edge.  The
    package's recursive restatement (rvo2.split_obstacles_kdtree, what the kernels are given) and the oracle's work-list one
    (crowd_oracle.split_obstacles) give the same vertex table on random scenes; the links stay consistent; a new vertex lies ON the edge it cuts,
    carries that edge's direction and is convex; the pieces of every polygon still close it with the original perimeter.  (UNPINNED: rvo2 absent.)"""
    from social_navigation_pyenvs_amd import rvo2

    rng = np.random.default_rng(3)
    splits = 0
    for trial in range(60):
        polys = []
        for _ in range(int(rng.integers(2, 6))):
            c = rng.uniform(-6, 6, 2); m = int(rng.integers(3, 7))
            ang = np.sort(rng.uniform(0, 2 * np.pi, m)); rad = rng.uniform(0.5, 2.5, m)
            polys.append((c + np.stack([np.cos(ang), np.sin(ang)], -1) * rad[:, None]).tolist())
        if trial % 7 == 0:
            polys.append([[-8.0, 0.3], [8.0, 0.3]])                       # a one-sided wall (two vertices) through the scene
        raw = rvo2.process_obstacles(polys, kdtree_split=False)
        a, b = rvo2.process_obstacles(polys), orc.process_obstacles(polys)
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a[: len(raw), [0, 1, 2, 3, 4]], raw[:, [0, 1, 2, 3, 4]])      # the polygons' own vertices are untouched
        n0 = len(raw)
        splits += len(a) - n0
        nxt, prv = a[:, 5].astype(int), a[:, 6].astype(int)
        assert np.array_equal(prv[nxt], np.arange(len(a))) and np.array_equal(nxt[prv], np.arange(len(a)))
        for k in range(n0, len(a)):
            assert a[k, 4] == 1.0
            # walk back / forward to the original vertices of the cut edge
            p = prv[k]
            while p >= n0:
                p = prv[p]
            q = int(raw[p, 5])
            e = raw[q, 0:2].astype(np.float64) - raw[p, 0:2]
            w = a[k, 0:2].astype(np.float64) - raw[p, 0:2]
            assert abs(e[0] * w[1] - e[1] * w[0]) / np.linalg.norm(e) < 1e-5 and -1e-6 <= (w @ e) / (e @ e) <= 1 + 1e-6
            np.testing.assert_array_equal(a[k, 2:4], raw[p, 2:4])
        # every cycle closes with its polygon's perimeter
        seen = np.zeros(len(a), bool)
        per_split, per_raw = 0.0, 0.0
        for s_ in range(len(a)):
            i = s_
            while not seen[i]:
                seen[i] = True
                per_split += float(np.linalg.norm(a[nxt[i], 0:2].astype(np.float64) - a[i, 0:2]))
                i = nxt[i]
        for i in range(n0):
            per_raw += float(np.linalg.norm(raw[int(raw[i, 5]), 0:2].astype(np.float64) - raw[i, 0:2]))
        assert abs(per_split - per_raw) < 1e-4 * per_raw
    assert splits > 50


def test_obstacle_pieces_give_the_edges_half_planes_except_at_piece_boundaries():
    """An agent walking along a long wall that processObstacles() cut into pieces (another polygon's edge line crosses it): its new velocity with
    the pieces equals the one with the uncut wall wherever the agent is not beside a cut (the pieces are collinear, convex joints: the same ORCA
    line, or two coinciding ones), and the restatement handles the cut vertices without artefacts (finite, inside the speed disc)."""
    from social_navigation_pyenvs_amd import rvo2

    rect = lambda x0, y0, x1, y1: [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]      # counter-clockwise
    # a long wall below the corridor, a pillar above it and a block at either end: the tree's root takes the pillar's side x = 0.5 as its
    # splitting line, which cuts the wall's bottom and top edges at (0.5, -1) and (0.5, 0)
    polys = [rect(-6.0, -1.0, 6.0, 0.0), rect(-0.5, 1.0, 0.5, 7.0), rect(-9.0, -4.0, -7.0, 4.0), rect(7.0, -4.0, 9.0, 4.0)]
    raw, cut = rvo2.process_obstacles(polys, kdtree_split=False), rvo2.process_obstacles(polys)
    assert len(cut) == len(raw) + 2 and [0.5, 0.0] in cut[len(raw):, 0:2].tolist()
    xs = np.linspace(-5.0, 5.0, 81)
    worst_far, near = 0.0, 0
    for x in xs:
        pos = np.array([[x, 0.45]]); vel = np.array([[0.6, -0.2]]); pref = np.array([[1.0, -0.5]])
        v0 = orc.orca_new_velocities_obst(pos, vel, pref, [0.3], [1.0], raw, time_step=0.25)[0]
        v1 = orc.orca_new_velocities_obst(pos, vel, pref, [0.3], [1.0], cut, time_step=0.25)[0]
        assert np.all(np.isfinite(v1)) and np.linalg.norm(v1) <= 1.0 + 1e-5
        if abs(x - 0.5) > 1.5:
            worst_far = max(worst_far, float(np.abs(v0 - v1).max()))
        else:
            near += 1
    assert worst_far < 1e-6 and near > 5, (worst_far, near)
