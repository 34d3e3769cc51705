"""the unicycle robot -- RobotAgent.step / compute_position with ActionRot(v, r) (robot_agent.py:119-136), applied once per
SUBSTEP by the Gym loop (social_nav_gym.py:240-245) -- against golden G17 (tests/golden/make_golden.py gen_g17_unicycle: that loop run directly
on the reference's objects; the reference's own step() cannot take an ActionRot, social_nav_sim.py:973).
CPU: the oracle (oracle/sfm_step.inc step_block, kinematics = 1) reproduces the recorded robot poses and crowd rows per substep in float64.
GPU: CS_ROBOT_UNICYCLE through every kernel that compiles the branch -- the shape-specialised 26-row build and the run-time robot build (a visible
robot: the crowd's view of it is checked through the humans' rows), the DPP row kernel and the plain 25-row build (an invisible robot), the grid path
(visible and invisible), ORCA and social-momentum crowds (the robot's motion does not depend on the crowd model) -- per substep of the fused launch,
plus the Gym head's swept test with a unicycle action."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from golden_io import load_cases  # noqa: E402

from oracle import crowd_oracle as orc  # noqa: E402

TWO_PI = 2.0 * math.pi


def unicycle_step(rb, v, r, dt):
    """robot_agent.py:119-136 in float64 on a [.., 13] safe-state row (a restatement local to this test: the expected robot record)"""
    rb = np.array(rb, dtype=np.float64, copy=True)
    rb[..., 0] += np.cos(rb[..., 2] + r) * v * dt
    rb[..., 1] += np.sin(rb[..., 2] + r) * v * dt
    rb[..., 2] = np.mod(rb[..., 2] + r, TWO_PI)
    rb[..., 3] = np.cos(rb[..., 2]) * v
    rb[..., 4] = np.sin(rb[..., 2]) * v
    return rb


def crowd_cases():
    return [c for c in load_cases("g17_unicycle") if not c.get("robot_only")]


def test_oracle_unicycle_robot_alone_g17():
    """RobotAgent.step(ActionRot) 200 times: the yaw wraps through 2 pi (python's % keeps it in [0, 2 pi)); oracle and the test's formula vs the reference"""
    cases = [c for c in load_cases("g17_unicycle") if c.get("robot_only")]
    assert len(cases) == 4
    P = np.zeros((1, 20)); P[0, :] = 1.0
    for c in cases:
        v, r = c["actions"][0]
        human = np.array([[50.0, 50.0, 0, 0, 0, 0, 0, 0, 0.3, 75.0, 50.0, 50.0, 1.0]])
        goals = np.array([[[50.0, 50.0]]])
        rb = c["robot0"].copy()
        mine = c["robot0"].copy()
        for k in range(c["n_substeps"]):
            _, _, rb = orc.step_block(0, human, goals, None, P, c["dt"], 1, np.zeros(1), True, robot=rb, action=(v, r), kinematics=1)
            mine = unicycle_step(mine, v, r, c["dt"])
            np.testing.assert_allclose(rb[:5], c["robots"][0, k, :5], rtol=0, atol=1e-12)
            np.testing.assert_allclose(mine[:5], c["robots"][0, k, :5], rtol=0, atol=1e-12)
        assert 0.0 <= c["robots"][0, :, 2].min() and c["robots"][0, :, 2].max() < TWO_PI
    assert any(np.any(np.abs(np.diff(c["robots"][0, :, 2])) > 3.0) for c in cases)      # the wrap really happens in the fixture


def test_oracle_unicycle_gym_loop_g17():
    """the substep loop of social_nav_gym.py:240-245 with a unicycle robot, crowd rows and robot pose per substep, float64"""
    cases = crowd_cases()
    assert len(cases) == 18 and {c["human_num"] for c in cases} == {5, 10, 25} and any(not c["robot_visible"] for c in cases)
    for ci, c in enumerate(cases):
        vis = bool(c["robot_visible"])
        S, goals, rb = c["states0"].copy(), c["goals0"].copy(), c["robot0"].copy()
        rp = (float(c["respawn_bounds"][0]), float(c["respawn_bounds"][1]), c["robot_safety_space"]) if c["respawn"] else (0.0, 0.0, 0.0)
        for k, (v, r) in enumerate(c["actions"]):
            for sub in range(c["n_substeps"]):
                S, goals, rb = orc.step_block(c["sfm_type"], S, goals, None, c["params"], c["dt"], 1, c["mm_safety"], c["all_params_equal"],
                                              robot_visible=vis, robot=rb, action=(v, r), kinematics=1, respawn=bool(c["respawn"]), respawn_par=rp)
                np.testing.assert_allclose(rb[:5], c["robots"][k, sub, :5], rtol=0, atol=1e-12, err_msg=f"case {ci} step {k} substep {sub}: robot")
                ref = c["states"][k, sub]
                n = c["human_num"]
                err = np.max(np.abs(S[:n, :8] - ref[:n, :8]) / np.maximum(1.0, np.abs(ref[:n, :8])))
                assert err < 1e-9, (ci, c["model"], c["scenario"], k, sub, err)
                np.testing.assert_allclose(goals, c["goals"][k, sub], rtol=0, atol=1e-9)
                if vis:   # the crowd's copy of the robot row (motion_model_manager.py:359) is the robot after ITS move
                    np.testing.assert_allclose(ref[n, :5], c["robots"][k, sub, :5], rtol=0, atol=0)
    # the fixture is not trivial: the robot turns, and a visible robot changes what the crowd does (same worlds with and without it differ)
    assert all(np.abs(np.diff(c["robots"][0, :, 2])).min() > 0.01 for c in cases)


def test_oracle_swept_collision_test_with_a_unicycle_action_g17():
    """collision_detection_and_reaching_goal + compute_reward_and_infos (social_nav_sim.py:949-1029) for an ActionRot, with the attribute the
    reference reads provided (robot.theta := robot.yaw, what DESIGN.md 5 says the build uses): oracle reward head vs the reference's values"""
    infos = set()
    for ci, c in enumerate(crowd_cases()):
        n = c["human_num"]
        for k, (v, r) in enumerate(c["actions"]):
            S_in = c["states0"] if k == 0 else c["states"][k - 1, -1]
            rb = c["robot0"] if k == 0 else c["robots"][k - 1, -1]
            vel = np.array([v * math.cos(r + rb[2]), v * math.sin(r + rb[2])])
            out = orc.collision_reward(S_in[:n, 0:2], S_in[:n, 3:5], S_in[:n, 8], rb[0:2], float(rb[8]), c["robot_goal"], vel, c["T"], 0.25 * k, 50.0)
            col, dmin, reach, reward, term, trunc = c["heads"][k]
            assert bool(out["collision"]) == bool(col) and bool(out["reaching_goal"]) == bool(reach), (ci, k)
            if not col:
                assert abs(out["dmin"] - dmin) < 1e-12
            assert abs(out["reward"] - reward) < 1e-12 and bool(out["terminated"]) == bool(term) and bool(out["truncated"]) == bool(trunc)
            assert out["info"] == c["infos"][k]
            infos.add(c["infos"][k])
    assert {"Nothing", "Danger", "Collision"} <= infos


# ================================================================================================================================ GPU
def _f32(a):
    return None if a is None else np.asarray(a, dtype=np.float32)


def _yaw_diff(a, b):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return np.minimum(d, TWO_PI - d)


@pytest.mark.gpu
def test_gpu_unicycle_gym_loop_g17_per_substep_and_against_the_reference():
    """Every G17 case through cs_step with CS_ROBOT_UNICYCLE: every substep of the fused launch against the f64 oracle from the GPU's own
    previous record (1e-5; robot records included), and the end of every Gym step against what the REFERENCE produced (the block bar of
    tests/test_gpu_facade.py: 5e-5 over 20 fused float32 substeps).  A visible robot is seen by the crowd: the humans' rows carry its effect."""
    from parity_util import fused_substeps_vs_oracle
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    variants = set()
    worst_block = 0.0
    for ci, c in enumerate(crowd_cases()):
        vis, n = bool(c["robot_visible"]), c["human_num"]
        rb_bounds = (float(c["respawn_bounds"][0]), float(c["respawn_bounds"][1])) if c["respawn"] else None
        S0 = _f32(c["states0"]); g0 = _f32(c["goals0"]); R0 = _f32(c["robot0"])
        for k, (v, r) in enumerate(c["actions"]):
            # re-synchronised to the reference before every Gym step (tests/test_gpu_facade.py's protocol)
            S_k = S0 if k == 0 else _f32(c["states"][k - 1, -1]); g_k = g0 if k == 0 else _f32(c["goals"][k - 1, -1]); R_k = R0 if k == 0 else _f32(c["robots"][k - 1, -1])
            if vis:
                S_k = S_k.copy(); S_k[n] = R_k
            cw = CrowdWorlds(S_k, g_k, _f32(c["params"]), _f32(c["mm_safety"]), None, type=c["sfm_type"], all_params_equal=c["all_params_equal"],
                             robot_row=vis, robot=R_k, respawn_bounds=rb_bounds)
            cw.unicycle = True
            variants.add(cw.step_variant().split(" grid")[0])
            A = np.array([[v, r]], np.float32)
            if vis:
                res = fused_substeps_vs_oracle(cw, c["sfm_type"], S_k[None], g_k[None], _f32(c["params"]), _f32(c["mm_safety"])[None], None, c["dt"], c["n_substeps"],
                                               c["all_params_equal"], respawn=bool(c["respawn"]), respawn_bounds=rb_bounds, robot_row=True, robot=R_k[None], action=A,
                                               kinematics=1, group="unicycle robot, visible: per substep inside the fused launch (G17 worlds)", what=f"G17 case {ci} step {k}")
                assert res["robot_records"] == c["n_substeps"]
            else:
                cw.step(c["dt"], c["n_substeps"], A)
            got, rob = cw.get_states()[0], cw.get_robot()[0]
            ref, rref = c["states"][k, -1], c["robots"][k, -1]
            # the robot after 20 unicycle substeps vs the reference: 20 float32 position increments, the yaw 20 float32 additions
            np.testing.assert_allclose(rob[[0, 1, 3, 4]], rref[[0, 1, 3, 4]], rtol=0, atol=1e-5)
            assert _yaw_diff(rob[2], rref[2]) < 5e-6
            # the bar of one Gym step (20 fused float32 substeps) as in tests/test_gpu_facade.py: 5e-5, or -- where 20 stiff substeps amplify float32
            # rounding beyond it (a respawned human at contact distance, Moussaid's sign(theta ~ 0)) -- 3 x what the float32 instantiation of the
            # ORACLE is off its float64 one over this very block
            up = lambda x: np.asarray(x, np.float32).astype(np.float64)
            rp = (rb_bounds[0], rb_bounds[1], c["robot_safety_space"]) if c["respawn"] else (0.0, 0.0, 0.0)
            args = (c["sfm_type"], up(S_k), up(g_k), None, up(c["params"]), c["dt"], c["n_substeps"], up(c["mm_safety"]), c["all_params_equal"])
            kw = dict(robot_visible=vis, robot=up(R_k), action=(float(np.float32(v)), float(np.float32(r))), kinematics=1, respawn=bool(c["respawn"]), respawn_par=rp)
            r64 = orc.step_block(*args, **kw)[0]
            with np.errstate(over="ignore", invalid="ignore"):
                r32 = orc.step_block(*args, dtype=np.float32, **kw)[0]
            e32 = float(np.max(np.abs(r32[:n][:, [0, 1, 3, 4]] - r64[:n][:, [0, 1, 3, 4]])))
            err = np.max(np.abs(got[:n][:, [0, 1, 3, 4]].astype(np.float64) - ref[:n][:, [0, 1, 3, 4]]))
            assert err < max(5e-5, 3.0 * e32), (ci, c["model"], c["scenario"], k, err, e32)
            worst_block = max(worst_block, err / max(5e-5, 3.0 * e32))
    # the builds this went through: the 26-row shape-specialised one, the run-time robot build (6 / 11 rows), the DPP row kernel and the plain
    # 25-row build (invisible robot)
    joined = " | ".join(sorted(variants))
    assert "ROWS_CT=26" in joined and "k_sfm_step_row16" in joined and "ROWS_CT=25" in joined, joined
    print("unicycle variants:", joined, "; worst block error / bar", worst_block)


@pytest.mark.gpu
@pytest.mark.parametrize("n,visible", [(25, True), (10, True), (5, True), (25, False), (10, False), (40, True)])
def test_gpu_unicycle_batched_worlds_per_substep(n, visible):
    """A batch of hybrid worlds (more than one wavefront, goal switches and respawns) with a unicycle robot per world, its own (v, r) each: every
    substep of the fused launch against the oracle, robot records checked; an invisible robot's 20 moves against the float64 formula."""
    from parity_util import fused_substeps_vs_oracle
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import SFMS, CrowdWorlds

    W = 3 * (64 // (n + int(visible))) + 1
    rng = np.random.default_rng(1700 + n + int(visible))
    for model in ("hsfm_farina", "sfm_guo", "hsfm_new_moussaid"):
        S, goals, P, rb = sc.hybrid_worlds(W, n, model, seed0=170 + n)
        rw = (np.arange(W) % 2 == 1).astype(np.int32)
        R = np.zeros((W, 13), np.float32)
        R[:, 0:2] = rng.uniform(-3, 3, (W, 2)); R[:, 2] = rng.uniform(-np.pi, 2 * np.pi, W); R[:, 8] = 0.3; R[:, 9] = 80; R[:, 10:12] = -R[:, 0:2]; R[:, 12] = 1.0
        A = np.stack([rng.uniform(0.2, 1.0, W), rng.uniform(-0.1, 0.1, W)], -1).astype(np.float32)
        if visible:
            S = np.concatenate([S, R[:, None, :]], axis=1)
        S32, g32, P32 = _f32(S), _f32(goals), _f32(P)
        cw = CrowdWorlds(S32, g32, P32, None, None, type=model, all_params_equal=True, respawn_bounds=rb, respawn_worlds=rw, robot_row=visible, robot=R)
        cw.unicycle = True
        for _ in range(2):          # two Gym steps in: distinct velocities (Moussaid's sign(theta ~ 0) at rest)
            cw.step(0.0125, 20, A)
        S_k, g_k, R_k = cw.get_states(), cw.get_goals(), cw.get_robot()
        if visible:
            res = fused_substeps_vs_oracle(cw, SFMS.index(model), S_k, g_k, P32, None, None, 0.0125, 20, True, respawn=rw, respawn_bounds=rb, robot_row=True, robot=R_k,
                                           action=A, kinematics=1, group="unicycle robot, visible: per substep inside the fused launch (hybrid batch)",
                                           what=f"{cw.step_variant().split(' grid')[0]} {model}")
            assert res["within"] >= res["substeps"] - res["ill_conditioned"] and res["robot_records"] == 20 * W, res
        else:
            cw.step(0.0125, 20, A)
        want = R_k.astype(np.float64)
        for _ in range(20):
            want = unicycle_step(want, A[:, 0].astype(np.float64), A[:, 1].astype(np.float64), float(np.float32(0.0125)))
        got = cw.get_robot()
        np.testing.assert_allclose(got[:, [0, 1, 3, 4]], want[:, [0, 1, 3, 4]], rtol=0, atol=1e-5)
        assert _yaw_diff(got[:, 2], want[:, 2]).max() < 5e-6 and np.all((got[:, 2] >= 0) & (got[:, 2] < TWO_PI + 1e-6))


@pytest.mark.gpu
@pytest.mark.parametrize("visible", [True, False])
def test_gpu_unicycle_grid_path(visible, monkeypatch):
    """Worlds beyond one block (bigworld.hip k_bw_robot): the same worlds through the grid path (forced by CROWDSTEP_BIGWORLD_MIN_ROWS) -- the
    robot's rows equal the one-block path's (the same float32 statements), the crowd agrees per substep with the oracle."""
    from parity_util import fused_substeps_vs_oracle
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import SFMS, CrowdWorlds

    n, W, model = 70, 3, "hsfm_farina"
    rng = np.random.default_rng(77)
    pos, yaw, goals = sc.circular_crossing(W, n, 9.0, 9)
    S, P = sc.make_states(pos, yaw, goals), np.tile(sc.default_params(model), (n, 1))
    R = np.zeros((W, 13), np.float32)
    R[:, 0:2] = rng.uniform(-2, 2, (W, 2)); R[:, 2] = rng.uniform(0, 6, W); R[:, 8] = 0.3; R[:, 9] = 80; R[:, 10:12] = -R[:, 0:2]; R[:, 12] = 1.0
    A = np.stack([rng.uniform(0.3, 1.0, W), rng.uniform(-0.1, 0.1, W)], -1).astype(np.float32)
    if visible:
        S = np.concatenate([S, R[:, None, :]], axis=1)
    monkeypatch.setenv("CROWDSTEP_BIGWORLD_MIN_ROWS", "32")
    cw = CrowdWorlds(_f32(S), _f32(goals), _f32(P), None, None, type=model, all_params_equal=True, robot_row=visible, robot=R)
    cw.unicycle = True
    assert "bw" in cw.step_variant() or "grid" in cw.step_variant().lower(), cw.step_variant()
    if visible:
        res = fused_substeps_vs_oracle(cw, SFMS.index(model), _f32(S), _f32(goals), _f32(P), None, None, 0.0125, 20, True, robot_row=True, robot=R, action=A, kinematics=1,
                                       group="unicycle robot, visible: grid path per substep", what="grid path unicycle")
        assert res["robot_records"] == 20 * W
    else:
        cw.step(0.0125, 20, A)
    want = R.astype(np.float64)
    for _ in range(20):
        want = unicycle_step(want, A[:, 0].astype(np.float64), A[:, 1].astype(np.float64), float(np.float32(0.0125)))
    got = cw.get_robot()
    np.testing.assert_allclose(got[:, [0, 1, 3, 4]], want[:, [0, 1, 3, 4]], rtol=0, atol=1e-5)
    assert _yaw_diff(got[:, 2], want[:, 2]).max() < 5e-6


@pytest.mark.gpu
def test_gpu_gym_head_swept_test_with_a_unicycle_action_g17():
    """cs_collision_reward with CS_ROBOT_UNICYCLE: the action rows are (v, r), the robot's velocity over the horizon is v (cos, sin)(r + yaw)
    (social_nav_sim.py:968-973 with robot.theta := robot.yaw): the reference's recorded heads.  (The head INSIDE the step launch, cs_gym_step, equals
    this one bit for bit with ActionRot rows: tests/test_gpu_generators.py::test_gym_step_is_the_head_and_the_body_in_one_launch.)"""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    codes = {"Nothing": 0, "Danger": 1, "ReachGoal": 2, "Collision": 3, "Timeout": 4}
    seen, differs = set(), 0
    for ci, c in enumerate(crowd_cases()):
        n, vis = c["human_num"], bool(c["robot_visible"])
        for k, (v, r) in enumerate(c["actions"]):
            S_in = c["states0"] if k == 0 else c["states"][k - 1, -1]
            rb = (c["robot0"] if k == 0 else c["robots"][k - 1, -1]).copy()
            rb[10:12] = c["robot_goal"]
            S_k = _f32(S_in).copy()
            if vis:
                S_k[n] = _f32(rb)
            cw = CrowdWorlds(S_k, _f32(c["goals0"]), _f32(c["params"]), _f32(c["mm_safety"]), None, type=c["sfm_type"], all_params_equal=c["all_params_equal"],
                             robot_row=vis, robot=_f32(rb))
            cw.unicycle = True
            out = cw.collision_reward(np.array([[v, r]], np.float32), c["T"], 0.25 * k)[0]
            col, dmin, reach, reward, term, trunc = c["heads"][k]
            assert bool(out[0]) == bool(col) and bool(out[2]) == bool(reach) and bool(out[4]) == bool(term) and bool(out[5]) == bool(trunc), (ci, k, out, c["heads"][k])
            assert int(out[6]) == codes[c["infos"][k]]
            if not col:
                assert abs(out[1] - dmin) < 2e-6, (ci, k, out[1], dmin)
            assert abs(out[3] - reward) < 2e-6
            seen.add(c["infos"][k])
            # a holonomic head fed the same two numbers would answer another question: the fixture tells the two apart
            cw.unicycle = False
            other = cw.collision_reward(np.array([[v, r]], np.float32), c["T"], 0.25 * k)[0]
            cw.unicycle = True
            differs += int(bool(other[0]) != bool(col) or (not col and abs(other[1] - dmin) > 1e-4))
    assert {"Nothing", "Danger", "Collision"} <= seen and differs >= 10, (seen, differs)


@pytest.mark.gpu
@pytest.mark.parametrize("model,visible", [("orca", True), ("orca", False), ("social_momentum", True), ("social_momentum", False)])
def test_gpu_unicycle_robot_beside_orca_and_social_momentum_crowds(model, visible):
    """The robot's motion does not depend on the crowd's model (social_nav_gym.py:240-243 moves it before update_humans whatever the model is):
    ORCA and social-momentum crowds take CS_ROBOT_UNICYCLE too (round 5 refused it).  A unicycle substep IS a holonomic substep with the velocity
    v (cos, sin)(yaw + r) plus the yaw update (robot_agent.py:119-136), so the fused 20-substep launch with ActionRot rows must give what twenty
    one-substep launches with those twenty ActionXY rows give -- the crowd included (it sees the robot through its row) -- and the robot must follow
    the float64 formula."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 41, 12
    rng = np.random.default_rng(5 + int(visible))
    pos, yaw, g = sc.circular_crossing(W, n, 4.0, 171)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    if model == "orca":
        dd = g[:, :, 0] - S[:, :, 0:2]
        S[:, :, 5:7] = dd / np.linalg.norm(dd, axis=-1, keepdims=True)
    R = np.zeros((W, 13), np.float32)
    R[:, 0:2] = rng.uniform(-1.5, 1.5, (W, 2)); R[:, 2] = rng.uniform(-np.pi, 2 * np.pi, W); R[:, 8] = 0.3; R[:, 9] = 80; R[:, 10:12] = -R[:, 0:2]; R[:, 12] = 1.0
    A = np.stack([rng.uniform(0.3, 1.0, W), rng.uniform(-0.12, 0.12, W)], -1).astype(np.float32)
    St = np.concatenate([S, R[:, None, :]], axis=1) if visible else S
    margin = np.full((W, n + int(visible)), 0.01, np.float32)

    def mk():
        return CrowdWorlds(St, g, None, margin, None, type=model, robot_row=visible, robot=R)

    uni = mk(); uni.unicycle = True
    for _ in range(2):
        uni.step(0.0125, 20, A)
    hol = mk()
    rb = R.astype(np.float64)
    dt = float(np.float32(0.0125))
    for _ in range(40):
        h = rb[:, 2] + A[:, 1].astype(np.float64)
        act = np.stack([A[:, 0] * np.cos(h), A[:, 0] * np.sin(h)], -1).astype(np.float32)
        hol.step(0.0125, 1, act)
        rb = unicycle_step(rb, A[:, 0].astype(np.float64), A[:, 1].astype(np.float64), dt)
    got, ref = uni.get_states(), hol.get_states()
    robot = uni.get_robot()
    np.testing.assert_allclose(robot[:, [0, 1, 3, 4]], rb[:, [0, 1, 3, 4]], rtol=0, atol=2e-5)     # 40 float32 increments
    assert _yaw_diff(robot[:, 2], rb[:, 2]).max() < 1e-5 and np.all((robot[:, 2] >= 0) & (robot[:, 2] < TWO_PI + 1e-6))
    np.testing.assert_allclose(hol.get_robot()[:, [0, 1, 3, 4]], robot[:, [0, 1, 3, 4]], rtol=0, atol=2e-5)
    err = np.abs(got[:, :n][..., [0, 1, 3, 4]].astype(np.float64) - ref[:, :n][..., [0, 1, 3, 4]]).max(axis=-1)
    # the two runs hand the crowd robot rows that differ by the host's vs the device's sin / cos (1e-7): both models are discontinuous in
    # their inputs (linear programmes; an arg-max over 20 actions), so a few humans next to a decision edge may part over 40 substeps
    assert np.mean(err < 1e-5) > 0.97, (model, visible, float(np.mean(err < 1e-5)), float(err.max()))
    if visible:   # ... and the crowd really reacts to the robot: without it the humans end elsewhere
        lone = CrowdWorlds(S, g, None, margin[:, :n], None, type=model)
        for _ in range(2):
            lone.step(0.0125, 20)
        assert np.abs(lone.get_states()[..., 0:2] - got[:, :n, 0:2]).max() > 1e-2


@pytest.mark.gpu
def test_gpu_facade_step_with_action_rot_g17():
    """The drop-in facade: SocialNavGym.step(ActionRot(v, r)) of a robot with kinematics = 'unicycle' -- what the reference's own step() would
    return if its :973 could run -- against G17: reward / flags / info of the head, the observation after the block, the robot's pose."""
    from test_facade_cpu import make_env

    from social_navigation_pyenvs_amd.crowd_nav.utils.action import ActionRot

    checked = 0
    for ci, c in enumerate(crowd_cases()):
        if c["rep"] != 0 or c["model"].endswith("moussaid"):      # (rep 1 moved the robot by hand, Moussaid was warmed up: states the facade's reset does not give)
            continue
        env = make_env(c["model"], c["scenario"], c["human_num"], c["robot_visible"])
        env.robot.kinematics = "unicycle"
        env.robot.policy.kinematics = "unicycle"
        if c["safety_space"] > 0:
            env.set_safety_space(c["safety_space"])
        env.reset(phase="test", test_case=c["test_case"])
        n = c["human_num"]
        np.testing.assert_allclose(env.motion_model_manager.states[:n, :8], c["states0"][:n, :8], rtol=0, atol=1e-12)
        for k, (v, r) in enumerate(c["actions"]):
            if k > 0:      # re-synchronise with the reference (per-step parity, tests/test_gpu_facade.py)
                mm = env.motion_model_manager
                mm.states[...] = c["states"][k - 1, -1]; mm.goals[...] = c["goals"][k - 1, -1]
                mm._sync_goal_lists_from_array()
                rs = c["robots"][k - 1, -1]
                env.robot.position = rs[0:2].copy(); env.robot.yaw = float(rs[2]); env.robot.linear_velocity = rs[3:5].copy()
                env.global_time = c["global_time0"] + 0.25 * k
            ob, reward, term, trunc, info = env.step(ActionRot(float(v), float(r)))
            col, dmin, reach, rew, t0, t1 = c["heads"][k]
            assert abs(reward - rew) < 1e-9 and (bool(term), bool(trunc)) == (bool(t0), bool(t1)) and type(info[0]).__name__ == c["infos"][k], (ci, k)
            rref = c["robots"][k, -1]
            np.testing.assert_allclose([*env.robot.position, *env.robot.linear_velocity], rref[[0, 1, 3, 4]], rtol=0, atol=1e-5)
            assert _yaw_diff(env.robot.yaw, rref[2]) < 5e-6
            got = np.array([[o.px, o.py, o.vx, o.vy] for o in ob])
            ref = c["states"][k, -1][:n][:, [0, 1, 3, 4]]
            assert np.abs(got - ref).max() < (5e-4 if c["respawn"] else 5e-5), (ci, k, np.abs(got - ref).max())
            checked += 1
    assert checked >= 20
