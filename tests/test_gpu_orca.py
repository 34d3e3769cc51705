"""GPU: the ORCA HIP kernel (through cs_step / cs_peek with type = CS_ORCA) against the C restatement.
Parity with the reference itself is UNPINNED for ORCA (rvo2 absent): see oracle/orca_oracle.c."""
import numpy as np
import pytest

from oracle import crowd_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def exact_orca_arithmetic(monkeypatch):
    """This module is the BIT-IDENTITY suite: the register-resident ORCA build in the library's DEFAULT arithmetic (cs_worlds.orca_math =
    CS_ORCA_MATH_DEFAULT -> exact: correctly rounded divide / square root, no FMA contraction) must equal oracle/orca_oracle.c bit for bit.
    The opt-in fast / fma arithmetic is measured per substep in tests/test_gpu_orca_fast.py.  (The default follows CROWDSTEP_ORCA_MATH, read
    once per process: the suite refuses to run under a non-exact one rather than silently testing another build.)"""
    import os

    from social_navigation_pyenvs_amd import _lib

    assert os.environ.get("CROWDSTEP_ORCA_MATH", "exact") == "exact" and _lib.load().cs_orca_default_math() == _lib.CS_ORCA_MATH_EXACT
    yield


def make_worlds(rng, W, n, robot=False, traffic=False):
    rows = n + int(robot)
    S = np.zeros((W, rows, 13), np.float32)
    goals = np.full((W, n, 2, 2), np.nan, np.float32)
    for w in range(W):
        if traffic:
            pts = []
            while len(pts) < rows:
                p = np.array([rng.uniform(-6.5, 6.5), rng.uniform(-1.5, 1.5)])
                if all(np.linalg.norm(p - q) > 0.75 for q in pts):
                    pts.append(p)
            pos = np.array(pts)
            goals[w, :, 0, 0] = -10.0
            goals[w, :, 0, 1] = pos[:n, 1]
        else:
            ang = np.sort(rng.uniform(0, 2 * np.pi, rows))
            pos = (3.0 + rng.uniform(-0.3, 0.3, rows))[:, None] * np.stack([np.cos(ang), np.sin(ang)], -1)
            goals[w, :, 0] = -pos[:n]
            goals[w, :, 1] = pos[:n]
        S[w, :, 0:2] = pos
        S[w, :, 3:5] = rng.normal(0, 0.3, (rows, 2))
        S[w, :, 8] = rng.uniform(0.25, 0.4, rows)
        S[w, :, 12] = rng.uniform(0.8, 1.2, rows)
        d = goals[w, :, 0] - pos[:n]
        nrm = np.linalg.norm(d, axis=1, keepdims=True)
        S[w, :n, 5:7] = np.where(nrm > S[w, :n, 12:13], d / nrm, d)
        S[w, :n, 10:12] = goals[w, :, 0]
    return S, goals


@pytest.mark.parametrize("n,robot,traffic", [(5, False, False), (10, True, False), (25, False, False), (25, True, True), (7, False, True)])
def test_orca_kernel_matches_restatement(n, robot, traffic):
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(100 + n)
    W = 37
    S, goals = make_worlds(rng, W, n, robot, traffic)
    rows = n + int(robot)
    margin = np.full((W, rows), 0.01, np.float32)
    robots = None
    action = None
    if robot:
        robots = S[:, -1].copy()
        action = rng.normal(0, 0.5, (W, 2)).astype(np.float32)
    bounds = (7.0, 1.5) if traffic else None
    cw = CrowdWorlds(S, goals, None, margin, None, type="orca", robot_row=robot, robot=robots, respawn_bounds=bounds)
    nsub = 20
    cw.step(0.0125, nsub, action)
    got = cw.get_states()
    ref, rgoals, rrobot = orc.orca_step_block(S, goals, margin, 0.0125, nsub, robot_visible=robot, robot=robots, action=action,
                                              respawn=traffic, bounds=bounds or (0, 0))
    # same float32 algorithm, IEEE ops, no contraction on either side -> agreement to rounding noise
    assert np.max(np.abs(got[..., [0, 1, 3, 4, 5, 6, 10, 11]] - ref[..., [0, 1, 3, 4, 5, 6, 10, 11]])) < 1e-5
    assert np.nanmax(np.abs(cw.get_goals() - rgoals)) < 1e-5
    if robot:
        np.testing.assert_allclose(cw.get_robot()[:, [0, 1, 3, 4]], rrobot[:, [0, 1, 3, 4]], atol=1e-6)
    # physical sanity: speeds within max speed.  With overlapping agents the collision branch uses
    # 1/timeStep = 80: ORCA line points sit ~25 m/s away and RVO2's float32 line-circle intersection
    # (disc = dot^2 + r^2 - |point|^2) cancels catastrophically -> overshoot up to ~2e-2 (same in the oracle)
    assert np.all(np.linalg.norm(got[:, :n, 3:5], axis=-1) <= S[:, :n, 12] + 5e-2)


def test_orca_register_lp3_bit_exact_on_a_circular_crossing():
    """25-agent circular crossing, the cfg4 workload: about a third of the agents have an infeasible programme in every
    substep, so this walks the register-resident linearProgram3 (lp3_fast10) for 160 substeps; same IEEE operations in the
    same order as the C restatement -> identical bits."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 24, 25
    pos, yaw, g = sc.circular_crossing(W, n, 7.0, 4242)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    d = g[:, :, 0] - S[:, :, 0:2]
    S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    margin = np.full((W, n), 0.01, np.float32)
    cw = CrowdWorlds(S, g, None, margin, None, type="orca")
    ref, rg = S, g
    used_lp3 = 0
    for _ in range(8):
        cw.step(0.0125, 20)
        ref, rg, _ = orc.orca_step_block(ref, rg, margin, 0.0125, 20)
        v, lines, nl = orc.orca_new_velocities(ref[0, :, 0:2], ref[0, :, 3:5], ref[0, :, 5:7], ref[0, :, 8] + 0.01, ref[0, :, 12],
                                               time_step=0.0125, return_lines=True)
        for a in range(n):
            L = lines[a, :nl[a]].astype(np.float64)
            used_lp3 += int(len(L) and np.max(L[:, 2] * (L[:, 1] - v[a, 1]) - L[:, 3] * (L[:, 0] - v[a, 0])) > 1e-6)
    np.testing.assert_array_equal(cw.get_states()[..., [0, 1, 3, 4, 5, 6]], ref[..., [0, 1, 3, 4, 5, 6]])
    assert used_lp3 >= 20, used_lp3   # the scene does exercise linearProgram3


@pytest.mark.parametrize("n,robot", [(65, False), (150, True), (256, False), (300, True), (512, False)])
def test_orca_worlds_of_more_than_64_rows(n, robot):
    """Above 64 rows a world takes a block of 256 / 512 lanes (several wavefronts: block-wide barriers and respawn vote)."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    n = n - int(robot)
    rng = np.random.default_rng(n)
    W, rows = 3, n + int(robot)
    S = np.zeros((W, rows, 13), np.float32)
    goals = np.full((W, n, 2, 2), np.nan, np.float32)
    side = int(np.ceil(np.sqrt(rows)))
    for w in range(W):   # a jittered lattice walking towards the mirrored position: dense, every agent has 10 neighbours
        gx, gy = np.meshgrid(np.arange(side), np.arange(side))
        pos = (np.stack([gx.ravel(), gy.ravel()], -1)[:rows] - side / 2) * 0.9 + rng.uniform(-0.1, 0.1, (rows, 2))
        S[w, :, 0:2] = pos
        S[w, :, 3:5] = rng.normal(0, 0.3, (rows, 2))
        S[w, :, 8] = rng.uniform(0.25, 0.35, rows)
        S[w, :, 12] = rng.uniform(0.8, 1.2, rows)
        goals[w, :, 0] = -pos[:n]
        goals[w, :, 1] = pos[:n]
        d = goals[w, :, 0] - pos[:n]
        nrm = np.linalg.norm(d, axis=1, keepdims=True)
        S[w, :n, 5:7] = np.where(nrm > S[w, :n, 12:13], d / nrm, d)
        S[w, :n, 10:12] = goals[w, :, 0]
    margin = np.full((W, rows), 0.01, np.float32)
    robots = S[:, -1].copy() if robot else None
    action = rng.normal(0, 0.5, (W, 2)).astype(np.float32) if robot else None
    cw = CrowdWorlds(S, goals, None, margin, None, type="orca", robot_row=robot, robot=robots, respawn_bounds=(side * 0.3, 1.0))
    cw.step(0.0125, 12, action)
    ref, rgoals, rrobot = orc.orca_step_block(S, goals, margin, 0.0125, 12, robot_visible=robot, robot=robots, action=action,
                                              respawn=True, bounds=(side * 0.3, 1.0))
    cols = [0, 1, 3, 4, 5, 6, 10, 11]
    np.testing.assert_array_equal(cw.get_states()[..., cols], ref[..., cols])
    np.testing.assert_array_equal(cw.get_goals(), rgoals)
    if robot:
        np.testing.assert_array_equal(cw.get_robot()[:, [0, 1, 3, 4]], rrobot[:, [0, 1, 3, 4]])


def test_orca_large_worlds_have_no_limits():
    """Beyond 512 rows -- or when the generic build's per-agent columns outgrow a block's LDS -- a world takes the grid path, which
    builds everything the one-block kernel does: robot row, respawn, static obstacles, any maxNeighbors (the parity of each:
    test_orca_grid_path_robot_row_obstacles_and_other_neighbour_counts)."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    S = np.zeros((1, 513, 13), np.float32)
    S[0, :, 0] = np.arange(513)
    S[0, :, 8] = 0.3
    S[0, :, 12] = 1.0
    g = np.zeros((1, 513, 1, 2), np.float32)
    CrowdWorlds(S, g, None, np.zeros((1, 513), np.float32), None, type="orca").step(0.0125, 1)
    CrowdWorlds(S, g, None, np.zeros((1, 513), np.float32), None, type="orca", respawn_bounds=(7.0, 1.5)).step(0.0125, 1)
    cw = CrowdWorlds(S, g[:, :512], None, np.zeros((1, 513), np.float32), None, type="orca", robot_row=True, robot=S[:, -1].copy())
    assert "k_bw_orca_step<FAST10=1>" in cw.step_variant(), cw.step_variant()
    cw.step(0.0125, 1, np.array([[0.5, 0.0]], np.float32))
    # the generic variant keeps (K + obstacle lines) x 40 B per agent in the LDS: 300 agents with a square of walls do not fit a block
    verts = orc.process_obstacles([[[-50, -50], [50, -50], [50, 50], [-50, 50]]])
    cw = CrowdWorlds(S[:, :300], g[:, :300], None, np.zeros((1, 300), np.float32), None, type="orca", orca_vertices=verts)
    assert "k_bw_orca_step<FAST10=0>" in cw.step_variant(), cw.step_variant()
    cw.step(0.0125, 1)


def test_orca_peek_does_not_commit():
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(5)
    S, goals = make_worlds(rng, 9, 6)
    margin = np.full((9, 6), 0.01, np.float32)
    cw = CrowdWorlds(S, goals, None, margin, None, type="orca")
    nxt = cw.peek(0.25)
    ref, _, _ = orc.orca_step_block(S, goals, margin, 0.25, 1)
    assert np.max(np.abs(nxt[..., [0, 1, 3, 4]] - ref[..., [0, 1, 3, 4]])) < 1e-5
    np.testing.assert_array_equal(cw.get_states(), S)
    np.testing.assert_array_equal(cw.get_goals(), goals)


def test_orca_rejects_update_humans_parallel():
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(6)
    S, goals = make_worlds(rng, 2, 5)
    cw = CrowdWorlds(S, goals, None, None, None, type="orca")
    with pytest.raises(ValueError):  # the reference raises ValueError for type > 8 (forces_parallel.py:211)
        cw.update_humans_parallel(0.0125)


def test_rvo2_module_drop_in():
    """``social_navigation_pyenvs_amd.rvo2`` used the way the reference uses Python-RVO2 (motion_model_manager.py:237-246,
    386-394; policy_no_train/orca.py:95-129): the positions / velocities after doStep equal the C restatement's."""
    from social_navigation_pyenvs_amd import rvo2

    rng = np.random.default_rng(7)
    n = 12
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    pos = 2.5 * np.stack([np.cos(ang), np.sin(ang)], -1) + rng.uniform(-0.2, 0.2, (n, 2))
    vel = rng.normal(0, 0.3, (n, 2))
    radius = rng.uniform(0.25, 0.4, n)
    vmax = rng.uniform(0.8, 1.2, n)
    sim = rvo2.PyRVOSimulator(1 / 60, 10, 10, 5, 5, 0.3, 1)
    ids = [sim.addAgent((pos[i, 0], pos[i, 1]), 10, 10, 5, 5, radius[i] + 0.01, vmax[i], (vel[i, 0], vel[i, 1])) for i in range(n)]
    assert ids == list(range(n)) and sim.getNumAgents() == n
    sim.processObstacles()   # no obstacle: a no-op, called unconditionally by the reference (:246)
    p32, v32 = pos.astype(np.float32), vel.astype(np.float32)
    for step in range(5):
        pref = -p32 / np.linalg.norm(p32, axis=1, keepdims=True)
        for i in range(n):
            sim.setAgentPrefVelocity(i, (float(pref[i, 0]), float(pref[i, 1])))
        sim.setTimeStep(0.25)
        sim.doStep()
        nv = orc.orca_new_velocities(p32, v32, pref, (radius + 0.01).astype(np.float32), vmax.astype(np.float32), 10.0, 10, 5.0, 0.25)
        p32 = p32 + nv * np.float32(0.25)
        v32 = nv
        got_p = np.array([sim.getAgentPosition(i) for i in range(n)])
        got_v = np.array([sim.getAgentVelocity(i) for i in range(n)])
        np.testing.assert_allclose(got_v, v32, rtol=0, atol=2e-5)
        np.testing.assert_allclose(got_p, p32, rtol=0, atol=2e-5)
        p32, v32 = got_p.astype(np.float32), got_v.astype(np.float32)   # re-synchronise (per-step parity)
    assert abs(sim.getGlobalTime() - 5 * 0.25) < 1e-12
    sim.setAgentPosition(0, (9.0, 9.0)); sim.setAgentVelocity(0, (0.0, 0.0)); sim.setAgentRadius(0, 0.5)
    assert sim.getAgentPosition(0) == (9.0, 9.0) and sim.getAgentRadius(0) == 0.5
    # per-agent neighborDist / maxNeighbors / timeHorizon, as RVO2 keeps them: see test_rvo2_module_per_agent_parameters
    k = sim.addAgent((0, 0), 5.0, 3, 2.0, 1.0)
    assert (sim.getAgentNeighborDist(k), sim.getAgentMaxNeighbors(k), sim.getAgentTimeHorizon(k), sim.getAgentTimeHorizonObst(k)) == (5.0, 3, 2.0, 1.0)
    assert sim.getAgentMaxNeighbors(0) == 10 and sim.getAgentTimeHorizon(0) == 5.0


def _scene_with_polygons(rng, W, n):
    polys = [[[-1.2, -0.4], [1.0, -0.6], [1.3, 0.5], [-0.9, 0.7]],                       # a box in the middle
             [[2.5, 2.0], [3.5, 2.2], [3.0, 3.2]],                                        # a triangle
             [[-3.5, -3.0], [-2.0, -3.0], [-2.0, -2.5], [-3.0, -2.5], [-3.0, -1.5], [-3.5, -1.5]]]  # an L (one reflex vertex)
    verts = orc.process_obstacles(polys)
    S = np.zeros((W, n, 13), np.float32)
    goals = np.full((W, n, 2, 2), np.nan, np.float32)

    def free(p, placed):
        if any(np.linalg.norm(p - q) < 0.75 for q in placed):
            return False
        for poly in polys:
            poly = np.array(poly)
            for i in range(len(poly)):
                a, b = poly[i], poly[(i + 1) % len(poly)]
                t = np.clip(np.dot(p - a, b - a) / np.dot(b - a, b - a), 0, 1)
                if np.linalg.norm(p - (a + t * (b - a))) < 0.45:
                    return False
            inside = False                         # even-odd ray casting (the L shape is not convex)
            for i in range(len(poly)):
                a, b = poly[i], poly[(i + 1) % len(poly)]
                if (a[1] > p[1]) != (b[1] > p[1]) and p[0] < a[0] + (p[1] - a[1]) * (b[0] - a[0]) / (b[1] - a[1]):
                    inside = not inside
            if inside:
                return False
        return True

    for w in range(W):
        placed = []
        while len(placed) < n:
            p = rng.uniform(-4, 4, 2)
            if free(p, placed):
                placed.append(p)
        pos = np.array(placed)
        S[w, :, 0:2] = pos
        goals[w, :, 0] = -pos
        goals[w, :, 1] = pos
    S[:, :, 3:5] = rng.normal(0, 0.3, (W, n, 2))
    S[:, :, 8] = rng.uniform(0.25, 0.35, (W, n))
    S[:, :, 12] = rng.uniform(0.8, 1.2, (W, n))
    d = goals[:, :, 0] - S[:, :, 0:2]
    S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    S[:, :, 10:12] = goals[:, :, 0]
    return S, goals, verts, polys


def test_orca_static_obstacles_kernel_matches_restatement():
    """ORCA worlds with polygon obstacles (obstacle ORCA lines + linearProgram3 with hard obstacle
    constraints), 20 fused substeps on the GPU against the C restatement, and nobody ends up inside a polygon."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(21)
    W, n = 40, 9
    S, goals, verts, polys = _scene_with_polygons(rng, W, n)
    margin = np.full((W, n), 0.01, np.float32)
    cw = CrowdWorlds(S, goals, None, margin, None, type="orca", orca_vertices=verts)
    ref_S, ref_g = S.copy(), goals.copy()
    worst = 0.0
    for block in range(6):                       # per-block parity from re-synchronised state
        cw.set_states(ref_S); cw.set_goals(ref_g)
        cw.step(0.05, 20)
        got = cw.get_states()
        ref_S, ref_g, _ = orc.orca_step_block(ref_S, ref_g, margin, 0.05, 20, verts=verts)
        err = np.max(np.abs(got[..., [0, 1, 3, 4]] - ref_S[..., [0, 1, 3, 4]]))
        worst = max(worst, err)
        assert err < 2e-4, (block, err)
        np.testing.assert_array_equal(cw.get_goals(), ref_g)
    p = ref_S[..., 0:2].reshape(-1, 2).astype(float)
    for poly in polys[:2]:                        # convex ones: strict inside test
        poly = np.array(poly)
        inside = np.ones(len(p), bool)
        for i in range(len(poly)):
            a, b = poly[i], poly[(i + 1) % len(poly)]
            inside &= (b[0] - a[0]) * (p[:, 1] - a[1]) - (b[1] - a[1]) * (p[:, 0] - a[0]) > 0
        assert not inside.any()
    moved = np.linalg.norm(ref_S[..., 0:2] - S[..., 0:2], axis=-1)
    assert np.median(moved) > 1.0                 # six seconds: the crowd did walk
    print("orca + obstacles: worst |err| GPU vs restatement over 6 blocks of 20 substeps:", worst)


def test_rvo2_module_with_obstacles():
    from social_navigation_pyenvs_amd import rvo2

    sim = rvo2.PyRVOSimulator(0.1, 10, 10, 5, 5, 0.31, 1.0)
    a = sim.addAgent((0.3, 2.0))
    assert sim.addObstacle([(6, 0), (-6, 0)]) == 0 and sim.getNumObstacleVertices() == 2
    sim.processObstacles()
    sim.setAgentVelocity(a, (0.0, -1.0))
    sim.setAgentPrefVelocity(a, (0.0, -1.0))
    sim.doStep()
    vx, vy = sim.getAgentVelocity(a)
    assert abs(vx) < 1e-6 and abs(vy + (2.0 - 0.31) / 5.0) < 1e-6    # the wall's cut-off line: v_y = -(d - r) / timeHorizonObst
    np.testing.assert_allclose(rvo2.process_obstacles([[[0, 0], [2, 0], [2, 2], [0, 2]]]), orc.process_obstacle([[0, 0], [2, 0], [2, 2], [0, 2]]))


def test_motion_model_manager_orca_with_walls():
    """MotionModelManager("orca", walls=[Obstacle...]): sim.addObstacle(list(wall.vertices)) + processObstacles
    (motion_model_manager.py:244-246) through the facade, one substep against the restatement."""
    from social_navigation_pyenvs_amd.social_gym.src.agent import HumanAgent, RobotAgent
    from social_navigation_pyenvs_amd.social_gym.src.motion_model_manager import MotionModelManager
    from social_navigation_pyenvs_amd.social_gym.src.obstacle import Obstacle

    box = [[-1.0, -0.5], [1.0, -0.5], [1.0, 0.5], [-1.0, 0.5]]
    starts = [(0.2, 1.6), (-0.6, -1.5), (2.0, 0.1), (-2.2, 0.3)]
    humans = [HumanAgent(None, i, "sfm_helbing", [x, y], 0.0, [[-x, -y], [x, y]], radius=0.3, mass=75, des_speed=1.0)
              for i, (x, y) in enumerate(starts)]
    mm = MotionModelManager("orca", False, False, humans, RobotAgent(None), [Obstacle(None, box)])
    S0 = mm.states.astype(np.float32).copy()
    g0 = mm.goals.astype(np.float32).copy()
    for _ in range(10):
        mm.update_humans(0.0, 0.1)
    ref, _, _ = orc.orca_step_block(S0[None], g0[None], np.full((1, 4), 0.01, np.float32), 0.1, 10,
                                    verts=orc.process_obstacles([box]))
    got = np.array([[*h.position, *h.linear_velocity] for h in humans])
    np.testing.assert_allclose(got, ref[0][:, [0, 1, 3, 4]], atol=2e-5)
    assert got[0, 1] > 0.5 + 0.3    # the human heading into the box from above is still outside it


def test_orca_cfg4_full_size():
    """BASELINE.json configs[3] at its full size: 4096 worlds x 25-agent ORCA circular crossing (R = 7, neighborDist 10,
    maxNeighbors 10, tau 5, radius 0.31, maxSpeed 1), three Gym steps of 20 fused substeps.  Size-independent properties:
    a permuted half-batch reproduces its worlds bit for bit (composition invariance), speeds respect maxSpeed; and 12 sampled
    worlds equal the C restatement bit for bit over the 60 substeps (the register-resident LP2 / LP3 build, k_orca_step<FAST10>)."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 4096, 25
    pos, yaw, g = sc.circular_crossing(W, n, 7.0, 1000)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    g = g.astype(np.float32)
    d = g[:, :, 0] - S[:, :, 0:2]
    S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    margin = np.full((W, n), 0.01, np.float32)

    def run(sel):
        cw = CrowdWorlds(S[sel], g[sel], None, margin[sel], None, type="orca", layout="soa")
        assert "k_orca_step<FAST10=1,MAXT=64>" in cw.step_variant() and "math=exact" in cw.step_variant(), cw.step_variant()
        for _ in range(3):
            cw.step(0.0125, 20)
        return cw.get_states(), cw.get_goals()

    full, gfull = run(np.arange(W))
    assert np.all(np.isfinite(full[..., :8]))
    rng = np.random.default_rng(4)
    sel = rng.permutation(W)[: W // 2 + 1]
    part, gpart = run(sel)
    np.testing.assert_array_equal(part, full[sel])
    np.testing.assert_array_equal(gpart, gfull[sel])
    assert np.all(np.linalg.norm(full[..., 3:5], axis=-1) <= S[..., 12] + 5e-2)
    sample = np.sort(rng.choice(W, 12, replace=False))
    ref, rg = S[sample], g[sample]
    for _ in range(3):
        ref, rg, _ = orc.orca_step_block(ref, rg, margin[sample], 0.0125, 20)
    cols = [0, 1, 3, 4, 5, 6, 10, 11]
    np.testing.assert_array_equal(full[sample][..., cols], ref[..., cols])
    np.testing.assert_array_equal(gfull[sample], rg)


def test_orca_dense_phase_soak_bit_identical():
    """A slice of tools/orca_soak.py inside the GPU suite: 256 worlds of the cfg4 crossing (25 agents, R = 7) over 700 substeps --
    out through the dense phase in which a third of the agents has an infeasible programme in every substep (linearProgram3 on
    16-lane rows) and back to the rim, goal switches included -- checked against the C restatement after EVERY Gym step (35
    blocks of 20 fused substeps), bit for bit: positions, velocities, preferred velocities, goal columns.  A world that ever
    differs fails the test.  (The restatement's own parity with rvo2 is unpinned, like all of ORCA.)"""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 256, 25
    pos, yaw, g = sc.circular_crossing(W, n, 7.0, 31337 + n)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    g = g.astype(np.float32)
    d = g[:, :, 0] - S[:, :, 0:2]
    S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    margin = np.full((W, n), 0.01, np.float32)
    cw = CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa")
    assert "k_orca_step<FAST10=1,MAXT=64>" in cw.step_variant(), cw.step_variant()
    ref, rg = S, g
    cols = [0, 1, 3, 4, 5, 6, 10, 11]
    closest = np.inf
    for block in range(35):
        cw.step(0.0125, 20)
        ref, rg, _ = orc.orca_step_block(ref, rg, margin, 0.0125, 20)
        got = cw.get_states()
        differ = np.any(got[..., cols] != ref[..., cols], axis=(1, 2))
        assert not differ.any(), f"Gym step {block + 1}: {int(differ.sum())} worlds differ from the restatement (first: {int(np.argmax(differ))})"
        dd = np.linalg.norm(ref[:, :, None, 0:2] - ref[:, None, :, 0:2], axis=-1) + 10.0 * np.eye(n)[None]
        closest = min(closest, float(np.median(dd.min(axis=(1, 2)))))
    np.testing.assert_array_equal(cw.get_goals(), rg)
    assert closest < 0.75                                  # the crowds really met (agents of radius 0.31 passing shoulder to shoulder)
    assert np.mean(np.linalg.norm(ref[..., 0:2] - S[..., 0:2], axis=-1)) > 5.0


def test_ieee_div_sqrt_sequences_are_correctly_rounded():
    """The ORCA kernels' 8-instruction divide and 9-instruction square root (the compiler's FMA sequences without their
    exponent-range handling) give the compiler's (IEEE, correctly rounded) results bit for bit on 2^31 random operand pairs of
    the range the linear programmes work in -- exact zeros included."""
    import ctypes as C

    from social_navigation_pyenvs_amd import _lib

    lib = _lib.load()
    out = (C.c_ulonglong * 4)()
    _lib.check(lib.cs_debug_divsqrt_check(C.c_ulonglong(1 << 31), C.c_uint(12345), out, C.c_void_p(None)))
    assert out[0] == 0 and out[1] == 0, (out[0], out[1], hex(out[2]), hex(out[3]))


def test_lp3_rows_equals_the_static_walk_bitwise():
    """linearProgram3 as one 16-lane row per (agent, violated line) (lp3_rows, the default) and as the statically unrolled
    per-lane walk (lp3_fast10, CROWDSTEP_ORCA_LP3=static) are the same arithmetic in two lane assignments: identical bits
    over 200 substeps of a dense 25-agent crossing, and of 10- and 40-agent worlds (3 and 1 worlds per wavefront)."""
    import os

    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    for n, R in ((25, 5.0), (10, 2.5), (40, 7.0)):
        W = 37
        pos, yaw, g = sc.circular_crossing(W, n, R, 99 + n)
        S = sc.make_states(pos, yaw, g).astype(np.float32)
        d = g[:, :, 0] - S[:, :, 0:2]
        S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
        margin = np.full((W, n), 0.01, np.float32)
        res = {}
        for mode in ("rows", "static"):
            os.environ["CROWDSTEP_ORCA_LP3"] = mode
            try:
                cw = CrowdWorlds(S, g, None, margin, None, type="orca")
                for _ in range(10):
                    cw.step(0.0125, 20)
                res[mode] = cw.get_states()
            finally:
                os.environ.pop("CROWDSTEP_ORCA_LP3", None)
        np.testing.assert_array_equal(res["rows"], res["static"])


def _lattice_world(W, rows, rng, spacing=0.9):
    S = np.zeros((W, rows, 13), np.float32)
    goals = np.full((W, rows, 2, 2), np.nan, np.float32)
    side = int(np.ceil(np.sqrt(rows)))
    for w in range(W):   # a jittered lattice walking towards the mirrored position: dense, every agent has 10 neighbours
        gx, gy = np.meshgrid(np.arange(side), np.arange(side))
        pos = (np.stack([gx.ravel(), gy.ravel()], -1)[:rows] - side / 2) * spacing + rng.uniform(-0.1, 0.1, (rows, 2))
        S[w, :, 0:2] = pos
        S[w, :, 3:5] = rng.normal(0, 0.3, (rows, 2))
        S[w, :, 8] = rng.uniform(0.25, 0.35, rows)
        S[w, :, 12] = rng.uniform(0.8, 1.2, rows)
        goals[w, :, 0] = -pos
        goals[w, :, 1] = pos
        d = goals[w, :, 0] - pos
        nrm = np.linalg.norm(d, axis=1, keepdims=True)
        S[w, :, 5:7] = np.where(nrm > S[w, :, 12:13], d / nrm, d)
        S[w, :, 10:12] = goals[w, :, 0]
    return S, goals


def test_orca_world_of_4096_agents_through_the_grid():
    """a world far beyond one block.  The agents are binned into cells of edge neighborDist (hashed buckets,
    counting sort per substep) and every agent walks the 3 x 3 cells around its own; the ten smallest (distSq, row) keys do not
    depend on the visiting order, so the world stays bit-identical to the restatement's index-order walk over all 4096 rows."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rows = 4096
    rng = np.random.default_rng(4096)
    S, goals = _lattice_world(1, rows, rng)
    margin = np.full((1, rows), 0.01, np.float32)
    cw = CrowdWorlds(S, goals, None, margin, None, type="orca")
    cw.step(0.0125, 6)
    ref, rgoals, _ = orc.orca_step_block(S, goals, margin, 0.0125, 6)
    cols = [0, 1, 3, 4, 5, 6, 10, 11]
    np.testing.assert_array_equal(cw.get_states()[..., cols], ref[..., cols])
    np.testing.assert_array_equal(cw.get_goals(), rgoals)
    assert np.max(np.abs(ref[..., 0:2] - S[..., 0:2])) > 0.01          # they did move


def test_orca_grid_path_respawn_and_peek_equal_the_restatement():
    """SURVEY.md §8 row f3 remainder for ORCA worlds on the grid path (forced onto small worlds): the parallel-traffic respawn rule
    (k_bw_respawn in its RVO2-agent form: plain radii, goal columns) and cs_peek (one step, nothing committed) -- bit-identical to
    the restatement over 3 x 20 substeps with several humans entering the 3 m zone, and to the one-block kernel's peek."""
    import os

    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 6, 40
    pos, yaw, g = sc.parallel_traffic(W, n, 30.0, 8.0, seed0=5)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    g = g.astype(np.float32)
    rng = np.random.default_rng(2)
    for w in range(W):
        for k, i in enumerate(rng.choice(n, 4, replace=False)):
            S[w, i, 0] = g[w, i, 0, 0] + 3.0 + 0.01 * (k + 1)              # on the edge of the respawn zone
    d = g[:, :, 0] - S[:, :, 0:2]
    S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    S[:, :, 10:12] = g[:, :, 0]
    margin = np.full((W, n), 0.01, np.float32)
    bounds = (15.0, 4.0)
    res = {}
    for mode in ("grid", "block"):
        if mode == "grid":
            os.environ["CROWDSTEP_BIGWORLD_MIN_ROWS"] = "1"
        try:
            cw = CrowdWorlds(S, g, None, margin, None, type="orca", respawn_bounds=bounds)
            before = cw.get_states().copy()
            pk = cw.peek(0.25)
            np.testing.assert_array_equal(cw.get_states(), before)
            np.testing.assert_array_equal(cw.get_goals(), g)
            for _ in range(3):
                cw.step(0.0125, 20)
            res[mode] = (cw.get_states(), cw.get_goals(), pk)
        finally:
            os.environ.pop("CROWDSTEP_BIGWORLD_MIN_ROWS", None)
    ref, rg = S, g
    for _ in range(3):
        ref, rg, _ = orc.orca_step_block(ref, rg, margin, 0.0125, 20, respawn=True, bounds=bounds)
    cols = [0, 1, 3, 4, 5, 6, 10, 11]
    for mode in ("grid", "block"):
        np.testing.assert_array_equal(res[mode][0][..., cols], ref[..., cols])
        np.testing.assert_array_equal(res[mode][1], rg)
    np.testing.assert_array_equal(res["grid"][2], res["block"][2])
    assert np.any(np.abs(ref[..., 0] - S[..., 0]) > 5.0)                   # somebody was respawned


@pytest.mark.parametrize("layout", ["aos", "soa"])
def test_orca_grid_path_on_small_worlds_equals_the_restatement(layout):
    """The grid path forced onto several small worlds at once (CROWDSTEP_BIGWORLD_MIN_ROWS): negative cell coordinates, buckets
    shared by several cells, agents that change cell between substeps, goal rotation -- bit-identical over 40 substeps."""
    import os

    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(7)
    W, rows = 5, 150
    S, goals = _lattice_world(W, rows, rng, spacing=3.5)               # 43 m across: several cells of edge 10 m
    S[:, ::11, 0:2] *= 0.05                                             # some agents start next to their goal: goal rotation
    goals[:, ::11, 0] = S[:, ::11, 0:2] + 0.2
    S[:, :, 10:12] = goals[:, :, 0]
    margin = np.full((W, rows), 0.01, np.float32)
    os.environ["CROWDSTEP_BIGWORLD_MIN_ROWS"] = "1"
    try:
        cw = CrowdWorlds(S, goals, None, margin, None, type="orca", layout=layout)
        for _ in range(2):
            cw.step(0.0125, 20)
        got, ggoals = cw.get_states(), cw.get_goals()
    finally:
        os.environ.pop("CROWDSTEP_BIGWORLD_MIN_ROWS", None)
    ref, rgoals, _ = orc.orca_step_block(S, goals, margin, 0.0125, 40)
    cols = [0, 1, 3, 4, 5, 6, 10, 11]
    np.testing.assert_array_equal(got[..., cols], ref[..., cols])
    np.testing.assert_array_equal(ggoals, rgoals)


@pytest.mark.parametrize("case", ["robot_action", "robot_respawn", "obstacles", "k5", "k16_obstacles_robot"])
def test_orca_grid_path_robot_row_obstacles_and_other_neighbour_counts(case):
    """SURVEY.md §8 row f3, the rest of the ORCA grid path (forced onto small worlds so that the restatement checks it quickly): the robot
    as the last row, moved by its action through cs_worlds.d_robot and seen one substep late (motion_model_manager.py:389), the respawn
    rule with the robot row in its maxima, static obstacles and maxNeighbors other than 10 (the generic LDS-column solve with its
    neighbour columns filled in (distSq, row) order) -- bit-identical to the C restatement and to the one-block kernel."""
    import os

    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(len(case))
    robot = "robot" in case
    verts = None
    kw = {}
    if "obstacles" in case:
        W, n = 6, 30
        S, goals, verts, _ = _scene_with_polygons(rng, W, n + int(robot))
        goals = goals[:, :n]
        S[:, n:, 5:7] = 0.0
        kw["verts"] = verts
    elif case == "robot_respawn":
        from social_navigation_pyenvs_amd import scenarios as sc
        W, n = 5, 40
        pos, yaw, g = sc.parallel_traffic(W, n + 1, 30.0, 8.0, seed0=9)
        S = sc.make_states(pos, yaw, g).astype(np.float32)
        goals = g[:, :n].astype(np.float32)
        for w in range(W):
            for k, i in enumerate(rng.choice(n, 4, replace=False)):
                S[w, i, 0] = goals[w, i, 0, 0] + 3.0 + 0.01 * (k + 1)
        S[:, n, 0] = 14.0                                               # the robot is the rightmost row: it sets the respawn abscissa
        d = goals[:, :, 0] - S[:, :n, 0:2]
        S[:, :n, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
        S[:, :n, 10:12] = goals[:, :, 0]
        kw.update(respawn=True, bounds=(15.0, 4.0))
    else:
        W, n = 5, 90
        S, goals = _lattice_world(W, n + int(robot), rng, spacing=1.1)
        goals = goals[:, :n]
    rows = n + int(robot)
    K = 5 if case == "k5" else (16 if case.startswith("k16") else 10)
    margin = np.full((W, rows), 0.01, np.float32)
    robots = S[:, -1].copy() if robot else None
    action = rng.normal(0, 0.5, (W, 2)).astype(np.float32) if robot else None
    res = {}
    for mode in ("grid", "block"):
        if mode == "grid":
            os.environ["CROWDSTEP_BIGWORLD_MIN_ROWS"] = "1"
        try:
            cw = CrowdWorlds(S, goals, None, margin, None, type="orca", robot_row=robot, robot=robots, orca_vertices=verts,
                             respawn_bounds=kw.get("bounds"))
            cw.orca_params["max_neighbors"] = K
            assert ("k_bw_orca_step" in cw.step_variant()) == (mode == "grid"), cw.step_variant()
            for _ in range(2):
                cw.step(0.0125, 20, action)
            res[mode] = (cw.get_states(), cw.get_goals(), cw.get_robot() if robot else None)
        finally:
            os.environ.pop("CROWDSTEP_BIGWORLD_MIN_ROWS", None)
    ref, rg, rr = S, goals, robots
    for _ in range(2):
        ref, rg, rr = orc.orca_step_block(ref, rg, margin, 0.0125, 20, robot_visible=robot, robot=rr, action=action, max_nb=K, **kw)
    cols = [0, 1, 3, 4, 5, 6, 10, 11]
    np.testing.assert_array_equal(res["grid"][0], res["block"][0])
    np.testing.assert_array_equal(res["grid"][1], res["block"][1])
    np.testing.assert_array_equal(res["grid"][0][:, :n][..., cols], ref[:, :n][..., cols])
    np.testing.assert_array_equal(res["grid"][1], rg)
    if robot:
        np.testing.assert_array_equal(res["grid"][2][:, [0, 1, 3, 4]], rr[:, [0, 1, 3, 4]])
        np.testing.assert_array_equal(res["grid"][0][:, n, [0, 1, 3, 4]], rr[:, [0, 1, 3, 4]])
    assert np.max(np.abs(ref[:, :n, 0:2] - S[:, :n, 0:2])) > 0.05


def test_rvo2_module_with_a_thousand_agents():
    """The PyRVOSimulator facade beyond one block: 1000 agents (the kernel's grid neighbour search), one doStep against the
    restatement's index-order walk, bit for bit."""
    from social_navigation_pyenvs_amd import rvo2

    rng = np.random.default_rng(1000)
    n = 1000
    S, goals = _lattice_world(1, n, rng)
    pos, vel, rad, vmax = S[0, :, 0:2], S[0, :, 3:5], S[0, :, 8], S[0, :, 12]
    pref = S[0, :, 5:7]
    sim = rvo2.PyRVOSimulator(1 / 60, 10, 10, 5, 5, 0.3, 1)
    for i in range(n):
        sim.addAgent((float(pos[i, 0]), float(pos[i, 1])), 10, 10, 5, 5, float(rad[i]) + 0.01, float(vmax[i]), (float(vel[i, 0]), float(vel[i, 1])))
        sim.setAgentPrefVelocity(i, (float(pref[i, 0]), float(pref[i, 1])))
    sim.setTimeStep(0.0125)
    sim.doStep()
    nv = orc.orca_new_velocities(pos, vel, pref, (rad + np.float32(0.01)).astype(np.float32), vmax, 10.0, 10, 5.0, 0.0125)
    got_v = np.array([sim.getAgentVelocity(i) for i in range(n)], dtype=np.float32)
    got_p = np.array([sim.getAgentPosition(i) for i in range(n)], dtype=np.float32)
    np.testing.assert_array_equal(got_v, nv)
    np.testing.assert_array_equal(got_p, pos + nv * np.float32(0.0125))


def _agent_params(rng, W, rows):
    ap = np.zeros((W, rows, 4), np.float32)
    ap[..., 0] = rng.choice([1.5, 3.0, 6.0, 10.0], (W, rows))          # neighborDist
    ap[..., 1] = rng.choice([0, 1, 3, 7, 10, 13], (W, rows))           # maxNeighbors (0: the agent ignores everybody)
    ap[..., 2] = rng.choice([0.5, 2.0, 5.0, 8.0], (W, rows))           # timeHorizon
    ap[..., 3] = rng.choice([0.5, 2.0, 5.0], (W, rows))                # timeHorizonObst
    return ap


@pytest.mark.parametrize("case", ["block", "obstacles_robot", "grid", "big_block"])
def test_orca_per_agent_rvo2_parameters_match_the_restatement(case):
    """RVO2 keeps neighborDist, maxNeighbors, timeHorizon, timeHorizonObst per agent (addAgent's arguments, motion_model_manager.py:241;
    the reference passes ORCA_DEFAULTS for everyone): cs_worlds.d_orca_agent_params on every ORCA path that can hold them -- the LDS-column
    kernel (one wavefront per block and one world per 256-lane block, with static obstacles and a robot row) and the grid path -- bit
    for bit against the C restatement with the same per-agent array, over fused substeps; and the parameters really matter (the
    trajectories differ from the uniform ones)."""
    import os

    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(len(case))
    W, n = (5, 20) if case != "big_block" else (2, 120)
    robot = case == "obstacles_robot"
    verts = None
    if case == "obstacles_robot":
        S, goals, verts, _ = _scene_with_polygons(rng, W, n)
        R = np.zeros((W, 13), np.float32); R[:, 0:2] = (4.5, 4.5); R[:, 8] = 0.3; R[:, 12] = 1.0
        S = np.concatenate([S, R[:, None, :]], axis=1)
    else:
        from social_navigation_pyenvs_amd import scenarios as sc
        pos, yaw, g = sc.circular_crossing(W, n, 4.0 if n <= 20 else 14.0, 31)
        S = sc.make_states(pos, yaw, g).astype(np.float32)
        d = g[:, :, 0] - S[:, :, 0:2]
        S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
        S[:, :, 3:5] = rng.normal(0, 0.3, (W, n, 2))
        goals = g.astype(np.float32)
    rows = n + int(robot)
    margin = np.full((W, rows), 0.01, np.float32)
    ap = _agent_params(rng, W, rows)
    res = {}
    if case == "grid":
        os.environ["CROWDSTEP_BIGWORLD_MIN_ROWS"] = "1"
    try:
        for key, params in (("per_agent", ap), ("uniform", None)):
            cw = CrowdWorlds(S, goals, None, margin, None, type="orca", robot_row=robot, robot=S[:, n] if robot else None, orca_vertices=verts,
                             orca_agent_params=params, layout="soa" if case == "grid" else "aos")
            v = cw.step_variant()
            assert ("k_bw_orca_step" in v) == (case == "grid"), v
            if params is not None:
                assert "FAST10=0" in v, v                                        # the register-resident ten-neighbour solve holds ONE parameter set
            cw.step(0.0125, 12)
            res[key] = (cw.get_states(), cw.get_goals())
    finally:
        os.environ.pop("CROWDSTEP_BIGWORLD_MIN_ROWS", None)
    ref, rgoals, _ = orc.orca_step_block(S, goals, margin, 0.0125, 12, robot_visible=robot, robot=S[:, n] if robot else None, verts=verts, max_nb=13,
                                         agent_params=ap)
    np.testing.assert_array_equal(res["per_agent"][0][:, :n, [0, 1, 3, 4, 5, 6]], ref[:, :n, [0, 1, 3, 4, 5, 6]])
    np.testing.assert_array_equal(np.nan_to_num(res["per_agent"][1]), np.nan_to_num(rgoals))
    assert np.max(np.abs(res["per_agent"][0][:, :n, 0:2] - res["uniform"][0][:, :n, 0:2])) > 1e-4
    # an agent with maxNeighbors = 0 sees nobody: it walks its preferred velocity (clipped to maxSpeed) whatever stands in its way
    blind = ap[..., 1] == 0
    blind[:, n:] = False
    assert blind.any()
    one = CrowdWorlds(S, goals, None, margin, None, type="orca", robot_row=robot, robot=S[:, n] if robot else None, orca_vertices=None,
                      orca_agent_params=ap)
    one.step(0.0125, 1)
    v1 = one.get_states()[..., 3:5]
    pref = S[..., 5:7] * np.minimum(1.0, S[..., 12:13] / np.maximum(np.linalg.norm(S[..., 5:7], axis=-1, keepdims=True), 1e-9))
    np.testing.assert_allclose(v1[blind], pref[blind], atol=1e-6)


def test_rvo2_module_per_agent_parameters():
    """The rvo2 facade with agents of different neighborDist / maxNeighbors / timeHorizon (addAgent's arguments and the setAgent* calls)
    == the C restatement with the same per-agent array, step by step; changing a parameter between steps takes effect."""
    from social_navigation_pyenvs_amd import rvo2

    rng = np.random.default_rng(3)
    n = 14
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    pos = 2.8 * np.stack([np.cos(ang), np.sin(ang)], -1) + rng.uniform(-0.2, 0.2, (n, 2))
    vel = rng.normal(0, 0.3, (n, 2))
    ap = _agent_params(rng, 1, n)[0]
    sim = rvo2.PyRVOSimulator(0.25, 10, 10, 5, 5, 0.3, 1)
    for i in range(n):
        sim.addAgent((pos[i, 0], pos[i, 1]), float(ap[i, 0]), int(ap[i, 1]), float(ap[i, 2]), float(ap[i, 3]), 0.31, 1.0, (vel[i, 0], vel[i, 1]))
    p32, v32 = pos.astype(np.float32), vel.astype(np.float32)
    S = np.zeros((1, n, 13), np.float32); S[0, :, 8] = 0.31; S[0, :, 12] = 1.0
    goals = np.full((1, n, 1, 2), 1.0e9, np.float32)
    for step in range(6):
        if step == 3:                                   # a parameter changed between two steps
            sim.setAgentMaxNeighbors(2, 0); sim.setAgentTimeHorizon(5, 0.5); sim.setAgentNeighborDist(7, 1.5)
            ap[2, 1] = 0; ap[5, 2] = 0.5; ap[7, 0] = 1.5
        pref = -p32 / np.linalg.norm(p32, axis=1, keepdims=True)
        for i in range(n):
            sim.setAgentPrefVelocity(i, (float(pref[i, 0]), float(pref[i, 1])))
        sim.doStep()
        S[0, :, 0:2], S[0, :, 3:5], S[0, :, 5:7] = p32, v32, pref
        ref, _, _ = orc.orca_step_block(S, goals, np.zeros((1, n), np.float32), 0.25, 1, max_nb=int(ap[:, 1].max()), neighbor_dist=float(ap[:, 0].max()),
                                        agent_params=ap[None])
        got_p = np.array([sim.getAgentPosition(i) for i in range(n)], np.float32)
        got_v = np.array([sim.getAgentVelocity(i) for i in range(n)], np.float32)
        np.testing.assert_array_equal(got_v, ref[0, :, 3:5])
        np.testing.assert_array_equal(got_p, ref[0, :, 0:2])
        p32, v32 = got_p, got_v


def test_orca_obstacle_pieces_of_the_kdtree_split_bit_identical():
    """processObstacles() cuts edges (RVO2's obstacle kd-tree, rvo2.split_obstacles_kdtree): a corridor scene whose long wall IS cut -- the
    package's vertex table equals the oracle's restatement of the split, and crowds walking along the cut wall step bit-identically to the C
    restatement on those records (one-block generic build and the grid path)."""
    import os

    from social_navigation_pyenvs_amd import rvo2
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rect = lambda x0, y0, x1, y1: [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]
    polys = [rect(-6.0, -1.0, 6.0, 0.0), rect(-0.5, 1.6, 0.5, 7.0), rect(-9.0, -4.0, -7.0, 4.0), rect(7.0, -4.0, 9.0, 4.0)]
    verts = rvo2.process_obstacles(polys)
    np.testing.assert_array_equal(verts, orc.process_obstacles(polys))
    assert len(verts) == 18 and len(rvo2.process_obstacles(polys, kdtree_split=False)) == 16
    rng = np.random.default_rng(8)
    W, n = 24, 9
    S = np.zeros((W, n, 13), np.float32)
    goals = np.full((W, n, 2, 2), np.nan, np.float32)
    for w in range(W):
        xs = np.linspace(-5.0, 5.0, n) + rng.uniform(-0.15, 0.15, n)
        S[w, :, 0] = xs; S[w, :, 1] = rng.uniform(0.40, 1.1, n)          # in the corridor between the wall's top edge and the pillar
        S[w, :, 3:5] = rng.normal(0, 0.3, (n, 2)); S[w, :, 8] = 0.3; S[w, :, 9] = 75; S[w, :, 12] = 1.0
        goals[w, :, 0, 0] = -xs; goals[w, :, 0, 1] = -0.2                    # goals below the wall's top edge: everybody pushes against it
        goals[w, :, 1] = S[w, :, 0:2]
        d = goals[w, :, 0] - S[w, :, 0:2]
        S[w, :, 5:7] = d / np.linalg.norm(d, axis=1, keepdims=True); S[w, :, 10:12] = goals[w, :, 0]
    margin = np.full((W, n), 0.01, np.float32)
    ref, rgoals, _ = orc.orca_step_block(S, goals, margin, 0.0125, 40, verts=verts)
    for grid in (False, True):
        if grid:
            os.environ["CROWDSTEP_BIGWORLD_MIN_ROWS"] = "4"
        try:
            cw = CrowdWorlds(S, goals, None, margin, None, type="orca", orca_vertices=verts)
            assert ("k_bw_orca_step" in cw.step_variant()) == grid, cw.step_variant()
            cw.step(0.0125, 40)
            got = cw.get_states()
        finally:
            os.environ.pop("CROWDSTEP_BIGWORLD_MIN_ROWS", None)
        np.testing.assert_array_equal(got[..., [0, 1, 3, 4, 5, 6, 10, 11]], ref[..., [0, 1, 3, 4, 5, 6, 10, 11]])
    assert np.abs(ref[..., 0:2] - S[..., 0:2]).max() > 0.2                      # they moved; nobody went through the wall
    assert ref[..., 1].min() > 0.25
