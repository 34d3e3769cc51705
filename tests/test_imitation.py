"""The robot driven by a human motion model (imitation learning; SURVEY.md §8b seam: set_human_motion_model_as_robot_policy +
imitation_learning_step).  Golden G11 was recorded from the reference (tests/golden/make_golden.py gen_g11_imitation)."""
import numpy as np
import pytest

from golden_io import load_cases
from oracle import crowd_oracle as orc

DT = 0.0125


def _walls_cases():
    return [c for c in load_cases("g11_imitation") if c["family"] == "walls"]


def test_oracle_robot_model_matches_reference_g11():
    """The numpy restatement of update_robot against the recorded substeps (float64: rounding noise only)."""
    worst = 0.0
    for c in _walls_cases():
        n, vis = c["n"], c["robot_visible"]
        type_ = orc.ROBOT_MODELS.index(c["model"])
        S, goals, row = c["mm_states"][0].copy(), c["mm_goals"][0].copy(), c["robots"][0].copy()
        for k in range(60):
            row = orc.robot_model_substep(row, c["robot_params"], c["robot_model"], S[:n, 0:2], S[:n, 3:5], S[:n, 8], c["human_safety"],
                                          c["walls"], DT)
            if vis:
                S[n, 0:8] = row[0:8]
            S, goals, _ = orc.step_block(type_, S, goals, c["walls"], c["mm_params"], DT, 1, c["mm_safety"], c["all_params_equal"],
                                         robot_visible=vis)
            worst = max(worst, np.max(np.abs(row - c["robots"][k + 1])))
        assert np.max(np.abs(S - c["mm_states"][3])) < 1e-9
    assert worst < 1e-9, worst


def _robot13(row):
    return np.asarray(row[:13], dtype=np.float32)


@pytest.mark.gpu
def test_robot_model_kernel_matches_reference_walls_g11():
    """CrowdWorlds.imitation_block (cs_robot_model_step + cs_step, 20 substeps) against the reference, block by block from
    the recorded state: float32 kernels vs the float64 reference."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    worst = 0.0
    for ci, c in enumerate(_walls_cases()):
        n, vis = c["n"], c["robot_visible"]
        moussaid = c["model"].endswith("moussaid") or c["robot_model"].endswith("moussaid")
        for blk in range(c["nsub"] // 20):
            S, goals, row = c["mm_states"][blk], c["mm_goals"][blk], c["robots"][20 * blk]
            cw = CrowdWorlds(S, goals, c["mm_params"], c["mm_safety"], c["walls"], type=c["model"],
                             all_params_equal=c["all_params_equal"], robot_row=vis, robot=_robot13(row))
            hm = np.zeros(cw.rows, np.float32)
            hm[:n] = c["human_safety"]
            cw.set_robot_model(c["robot_model"], c["robot_params"], float(row[13]), hm)
            cw.d_robot_memory.upload(np.asarray(row[14:16], dtype=np.float32).reshape(1, 2))
            cw.imitation_block(DT, 20)
            rb = cw.get_robot()[0]
            ref = c["robots"][20 * (blk + 1)]
            er = np.max(np.abs(rb[:8] - ref[:8]))
            eh = np.max(np.abs(cw.get_states()[0][:n, [0, 1, 3, 4]] - c["mm_states"][blk + 1][:n, [0, 1, 3, 4]]))
            tol = 2e-3 if moussaid else 5e-5   # Moussaid: sign(theta) discontinuity
            assert er < tol and eh < tol, (ci, c["robot_model"], c["model"], blk, er, eh)
            if not moussaid:
                worst = max(worst, er, eh)
            # desired force = mass * (direction * speed - velocity) / relaxation_time: 160 x the velocity error
            np.testing.assert_allclose(cw.d_robot_memory.download()[0], ref[14:16], rtol=0, atol=400 * tol)
    assert worst < 5e-5


@pytest.mark.gpu
def test_gym_imitation_learning_step_g11():
    """The Gym seam: set_human_motion_model_as_robot_policy + imitation_learning_step, re-synchronised with the reference
    before every step (trajectories are chaotic)."""
    from test_facade_cpu import make_env

    for ci, c in enumerate(load_cases("g11_imitation")):
        if c["family"] != "gym":
            continue
        env = make_env(c["model"], c["scenario"], c["human_num"], c["robot_visible"], c["headed_obs"])
        env.set_human_motion_model_as_robot_policy(c["robot_model"], False)
        if c["safety_space"] > 0:
            env.set_safety_space(c["safety_space"])
        env.reset(phase=c["phase"], test_case=c["test_case"])
        mm = env.motion_model_manager
        assert mm.robot_motion_model_title == c["robot_model"]       # kept across reset (social_nav_sim.py:174-179)
        np.testing.assert_allclose(mm.safety_space, c["mm_safety"], atol=1e-12)
        np.testing.assert_allclose(mm.states, c["mm_states"][0], atol=1e-12)
        moussaid = c["model"].endswith("moussaid") or c["robot_model"].endswith("moussaid")
        for k in range(len(c["rewards"])):
            mm.states[...] = c["mm_states"][k]
            mm.goals[...] = c["mm_goals"][k]
            mm._sync_goal_lists_from_array()
            r = c["robots"][k]
            env.robot.position, env.robot.yaw, env.robot.linear_velocity = r[0:2].copy(), float(r[2]), r[3:5].copy()
            env.robot.body_velocity, env.robot.angular_velocity = r[5:7].copy(), float(r[7])
            env.robot.desired_force = r[14:16].copy()
            assert abs(env.robot.safety_space - r[13]) < 1e-12
            env.global_time = 0.25 * k
            ob, reward, term, trunc, info = env.imitation_learning_step()
            if moussaid and k == 0:
                continue   # everybody at rest: the sign of theta is rounding noise in the reference
            tol = 2e-2 if moussaid else (3e-4 if c["respawn"] else 5e-5)
            ref = c["robots"][k + 1]
            got = np.array([*env.robot.position, env.robot.yaw, *env.robot.linear_velocity, *env.robot.body_velocity, env.robot.angular_velocity])
            assert np.max(np.abs(got - ref[:8])) < tol, (ci, c["robot_model"], c["model"], k, np.abs(got - ref[:8]))
            obs = np.array([[o.px, o.py, o.vx, o.vy] for o in ob])
            assert np.max(np.abs(obs - c["obs"][k + 1][:, :4])) < tol, (ci, k)
            assert (term, trunc) == (bool(c["terminated"][k]), bool(c["truncated"][k]))
            assert type(info[0]).__name__ == c["infos"][k]
            assert abs(reward - c["rewards"][k]) < 10 * tol
        assert abs(env.global_time - 0.25 * len(c["rewards"])) < 1e-9


def _orca_robot_oracle(S, n, robot, verts, human_margin, robot_margin, dt):
    """One doStep of the robot's own simulator (motion_model_manager.py:641-653) on the C restatement: humans with preferred
    velocity 0 and the robot last; only the robot's row is used."""
    pos = np.vstack([S[:n, 0:2], robot[None, 0:2]]).astype(np.float32)
    vel = np.vstack([S[:n, 3:5], robot[None, 3:5]]).astype(np.float32)
    pref = np.zeros_like(pos)
    d = robot[10:12] - robot[0:2]
    nrm = np.float32(np.sqrt(d[0] * d[0] + d[1] * d[1]))
    pref[n] = d / nrm if nrm > robot[12] else d
    radius = np.concatenate([S[:n, 8] + human_margin, [robot[8] + robot_margin]]).astype(np.float32)
    maxspeed = np.concatenate([S[:n, 12], [robot[12]]]).astype(np.float32)
    if verts is None:
        v = orc.orca_new_velocities(pos, vel, pref, radius, maxspeed, time_step=dt)
    else:
        v = orc.orca_new_velocities_obst(pos, vel, pref, radius, maxspeed, verts, time_step=dt)[0]
    out = robot.copy()
    out[3:5] = v[n]
    out[0:2] = robot[0:2] + v[n] * np.float32(dt)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("walls", [False, True])
def test_orca_robot_model_matches_restatement(walls):
    """robot_type = CS_ORCA among SFM humans: the robot's row of the C restatement, bit for bit, over 40 alternating
    substeps (parity with rvo2 itself is unpinned like all of ORCA)."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 6, 12
    rng = np.random.default_rng(5)
    pos, yaw, g = sc.circular_crossing(W, n, 4.0, 777)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    S[:, :, 3:5] = rng.normal(0, 0.3, (W, n, 2))
    P = np.tile(sc.default_params("sfm_helbing"), (n, 1))
    robot = np.zeros((W, 13), np.float32)
    robot[:, 0:2] = rng.uniform(-0.5, 0.5, (W, 2)) + [-3.0, 0.0]
    robot[:, 8], robot[:, 9], robot[:, 12] = 0.3, 80.0, 1.0
    robot[:, 10:12] = [3.5, 0.2]
    verts = orc.process_obstacles([[[-1.0, 1.2], [1.0, 1.2], [1.0, 1.6], [-1.0, 1.6]], [[0.5, -2.0], [1.5, -2.0], [1.0, -1.2]]]) if walls else None
    cw = CrowdWorlds(S, g, P, np.zeros((W, n), np.float32), None, type="sfm_helbing", all_params_equal=True, robot=robot)
    cw.set_robot_model("orca", None, 0.01 + 0.05, np.full((W, n), 0.06, np.float32), orca_vertices=verts)
    ref_robot = robot.copy()
    for k in range(40):
        Sg = cw.get_states()
        for w in range(W):
            ref_robot[w] = _orca_robot_oracle(Sg[w], n, ref_robot[w], verts, np.float32(0.06), np.float32(0.06), DT)
        cw.imitation_block(DT, 1)
        np.testing.assert_array_equal(cw.get_robot()[:, [0, 1, 3, 4]], ref_robot[:, [0, 1, 3, 4]])
    assert np.all(ref_robot[:, 0] > robot[:, 0] + 0.2)   # the robots did move towards their goals
    # update_robot(t, dt, just_velocities=True) for the ORCA robot (motion_model_manager.py:641-653): the doStep's new velocity is
    # kept, the robot's simulator agent is put back on robot.position -- the velocity of a full step, the position untouched
    before = cw.get_robot()
    Sg = cw.get_states()
    full = np.stack([_orca_robot_oracle(Sg[w], n, before[w], verts, np.float32(0.06), np.float32(0.06), DT) for w in range(W)])
    cw.robot_model_step(DT, just_velocities=True)
    after = cw.get_robot()
    np.testing.assert_array_equal(after[:, 3:5], full[:, 3:5])
    np.testing.assert_array_equal(after[:, 0:2], before[:, 0:2])
    assert np.any(full[:, 0:2] != before[:, 0:2])


@pytest.mark.gpu
def test_gym_imitation_with_orca_robot_and_orca_humans_runs():
    """train.config's default il_policy = orca: the robot reaches for its goal among ORCA humans, visible to them."""
    from test_facade_cpu import make_env

    env = make_env("orca", "circle_crossing", 5, True)
    env.set_human_motion_model_as_robot_policy("orca", False)
    env.set_safety_space(0.15)
    env.reset(phase="train", test_case=3)
    start = env.robot.position.copy()
    goal = np.array(env.robot.get_goal_position(), dtype=float)
    total, infos = 0.0, []
    for k in range(80):
        ob, reward, term, trunc, info = env.imitation_learning_step()
        total += reward
        infos.append(type(info[0]).__name__)
        mm = env.motion_model_manager
        n = len(env.humans)
        np.testing.assert_allclose(mm.states[n, 0:2], env.robot.position, atol=1e-6)   # set_state_orca(robot) after doStep
        if term or trunc:
            break
    assert infos[-1] == "ReachGoal", infos[-5:]
    assert np.linalg.norm(env.robot.position - goal) < env.robot.radius
    assert np.linalg.norm(start - goal) > 5.0


@pytest.mark.gpu
def test_batched_actual_collision_reward_matches_host_rule():
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(3)
    W, n = 64, 7
    S = np.zeros((W, n, 13), np.float32)
    S[:, :, 0:2] = rng.uniform(-2, 2, (W, n, 2))
    S[:, :, 8] = rng.uniform(0.2, 0.4, (W, n))
    robot = np.zeros((W, 13), np.float32)
    robot[:, 0:2] = rng.uniform(-2, 2, (W, 2))
    robot[:, 8] = 0.3
    robot[:, 10:12] = robot[:, 0:2] + rng.uniform(-0.4, 0.4, (W, 2))
    gt = rng.uniform(0, 52, W).astype(np.float32)
    cw = CrowdWorlds(S, np.zeros((W, n, 1, 2), np.float32), np.zeros((n, 20), np.float32), None, None, type=0, robot=robot)
    out = cw.actual_collision_reward(0.25, gt)
    seen = set()
    for w in range(W):
        d = np.sqrt(((S[w, :, 0:2] - robot[w, 0:2]) ** 2).sum(1)) - S[w, :, 8] - 0.3
        dmin = min(10000.0, d.min())
        coll, reach = dmin <= 0, np.linalg.norm(robot[w, 0:2] - robot[w, 10:12]) < 0.3
        if gt[w] >= 49.0:
            exp = (0.0, 0, 1, 4)
        elif coll:
            exp = (-0.25, 1, 0, 3)
        elif reach:
            exp = (1.0, 1, 0, 2)
        elif dmin < 0.2:
            exp = ((dmin - 0.2) * 0.5 * 0.25, 0, 0, 1)
        else:
            exp = (0.0, 0, 0, 0)
        seen.add(exp[3])
        assert abs(out[w, 3] - exp[0]) < 1e-6 and tuple(out[w, 4:7].astype(int)) == exp[1:], (w, out[w], exp)
        assert abs(out[w, 1] - dmin) < 1e-5 and bool(out[w, 0]) == coll and bool(out[w, 2]) == reach
    assert seen == {0, 1, 2, 3, 4}


@pytest.mark.gpu
@pytest.mark.parametrize("hmodel,rmodel,visible,safety", [("hsfm_new_guo", "sfm_helbing", True, 0.0), ("sfm_helbing", "hsfm_farina", False, 0.1),
                                                          ("orca", "orca", True, 0.15), ("sfm_guo", "orca", False, 0.0)])
def test_batched_imitation_matches_single_env(hmodel, rmodel, visible, safety):
    """BatchedSocialNavGym.imitation_learning_step == W single-world facades, step by step."""
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym
    from social_navigation_pyenvs_amd.social_gym.src.info import INFO_BY_CODE
    from test_facade_cpu import make_config, make_env

    W, steps = 5, 6
    cfg = make_config(hmodel, "circle_crossing", 5, visible)
    benv = BatchedSocialNavGym(cfg, W, robot_visible=visible)
    benv.reset(phase="val", first_case=11, safety_space=safety)
    benv.set_human_motion_model_as_robot_policy(rmodel, False, safety_space=safety)
    hist = [benv.imitation_learning_step() for _ in range(steps)]
    robots = benv.cw.get_robot()
    for w in range(W):
        env = make_env(hmodel, "circle_crossing", 5, visible)
        env.set_human_motion_model_as_robot_policy(rmodel, False)
        if safety > 0:
            env.set_safety_space(safety)
        env.reset(phase="val", test_case=11 + w)
        for k in range(steps):
            ob, r, t, tr, info = env.imitation_learning_step()
            obs, rew, term, trunc, code = hist[k]
            got = np.array([[o.px, o.py, o.vx, o.vy, o.radius] for o in ob])
            # the facade keeps float64 mirrors between steps, the batch stays in float32 on the device
            assert np.max(np.abs(got - obs[w])) < 2e-4, (w, k, np.max(np.abs(got - obs[w])))
            assert abs(r - rew[w]) < 1e-4 and t == bool(term[w]) and tr == bool(trunc[w])
            assert type(info[0]) is INFO_BY_CODE[int(code[w])] or isinstance(info[0], INFO_BY_CODE[int(code[w])])
        np.testing.assert_allclose(robots[w, [0, 1, 3, 4]], [*env.robot.position, *env.robot.linear_velocity], atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("n,rmodel", [(1, "sfm_guo"), (64, "hsfm_new_moussaid"), (150, "sfm_helbing"), (150, "hsfm_farina")])
def test_robot_model_kernel_edge_sizes_against_the_oracle(n, rmodel):
    """One human, a full wavefront of humans, and more humans than lanes (the lane-strided social-force sum)."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(n)
    side = int(np.ceil(np.sqrt(n + 1)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    pts = (np.stack([gx.ravel(), gy.ravel()], -1)[:n + 1] - side / 2) * 1.2 + rng.uniform(-0.2, 0.2, (n + 1, 2))
    S = np.zeros((1, n, 13), np.float32)
    S[0, :, 0:2] = pts[:n]
    S[0, :, 3:5] = rng.normal(0, 0.4, (n, 2))
    S[0, :, 8], S[0, :, 9], S[0, :, 12] = rng.uniform(0.25, 0.4, n), 75.0, 1.0
    goals = np.zeros((1, n, 1, 2), np.float32)
    goals[0, :, 0] = -S[0, :, 0:2]
    S[0, :, 10:12] = goals[0, :, 0]
    row = np.zeros(16)
    row[0:2], row[2], row[5:7], row[7] = pts[n], 0.4, [0.5, 0.1], 0.2
    row[3:5] = [0.5 * np.cos(0.4) - 0.1 * np.sin(0.4), 0.5 * np.sin(0.4) + 0.1 * np.cos(0.4)] if rmodel.startswith("hsfm") else [0.3, 0.2]
    row[8], row[9], row[10:12], row[12], row[13] = 0.3, 80.0, [side, side], 1.0, 0.02
    row = row.astype(np.float32).astype(np.float64)
    P = sc.default_params(rmodel)
    hs = np.full(n, 0.015, np.float32)
    # humans under ORCA so that they do not depend on SFM parameters; only the robot substep is compared
    cw = CrowdWorlds(S, goals, None, np.zeros((1, n), np.float32), None, type="orca", robot=_robot13(row))
    cw.set_robot_model(rmodel, P, 0.02, hs[None])
    cw.robot_model_step(DT)
    ref = orc.robot_model_substep(row, P.astype(np.float32).astype(np.float64), rmodel, S[0, :, 0:2].astype(np.float64),
                                  S[0, :, 3:5].astype(np.float64), S[0, :, 8].astype(np.float64), hs.astype(np.float64), None, DT)
    got = cw.get_robot()[0]
    # (2e-6 with IEEE divide / libm exp; the per-human loop now uses the crowd kernel's single-instruction rsq / exp: 1 ulp each)
    tol = 1e-3 if rmodel.endswith("moussaid") else 1e-5
    assert np.max(np.abs(got[:8] - ref[:8])) < tol * max(1.0, np.max(np.abs(ref[:8]))), (got[:8], ref[:8])
    # update_robot(t, dt, just_velocities=True) (motion_model_manager.py:72-85, 615-629): velocities integrate, the pose stays
    cw2 = CrowdWorlds(S, goals, None, np.zeros((1, n), np.float32), None, type="orca", robot=_robot13(row))
    cw2.set_robot_model(rmodel, P, 0.02, hs[None])
    cw2.robot_model_step(DT, just_velocities=True)
    refv = orc.robot_model_substep(row, P.astype(np.float32).astype(np.float64), rmodel, S[0, :, 0:2].astype(np.float64),
                                   S[0, :, 3:5].astype(np.float64), S[0, :, 8].astype(np.float64), hs.astype(np.float64), None, DT,
                                   just_velocities=True)
    gv = cw2.get_robot()[0]
    np.testing.assert_array_equal(gv[0:3], _robot13(row)[0:3])                                   # x, y, yaw untouched
    assert np.max(np.abs(gv[:8] - refv[:8])) < tol * max(1.0, np.max(np.abs(refv[:8]))), (gv[:8], refv[:8])
    if rmodel.startswith("hsfm"):
        np.testing.assert_array_equal(gv[5:8], got[5:8])                                         # body / angular velocity: same update


@pytest.mark.gpu
@pytest.mark.parametrize("hmodel,rmodel,n", [("hsfm_farina", "sfm_helbing", 25), ("sfm_guo", "hsfm_new_guo", 10), ("hsfm_new_moussaid", "hsfm_moussaid", 7)])
def test_imitation_block_two_launches_equal_the_alternating_launches(hmodel, rmodel, n):
    """cs_imitation_block with an invisible robot: the crowd's 20 fused substeps (recording what the robot sees at every substep)
    followed by the robot's 20 substeps == 20 x { cs_robot_model_step ; cs_step(1) } bit for bit, robot rows and crowd alike."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W = 37
    S, goals, P, rb = sc.hybrid_worlds(W, n, hmodel, seed0=5)
    rng = np.random.default_rng(n)
    robot = np.zeros((W, 13), np.float32)
    robot[:, 0:2] = rng.uniform(-3, 3, (W, 2)); robot[:, 8] = 0.3; robot[:, 9] = 80; robot[:, 10:12] = -robot[:, 0:2]; robot[:, 12] = 1.0
    walls = sc.polygon_walls()
    res = []
    for fused in (True, False):
        cw = CrowdWorlds(S, goals, P, None, walls, type=hmodel, all_params_equal=True, respawn_bounds=rb,
                         respawn_worlds=(np.arange(W) % 2 == 1).astype(np.int32), robot=robot)
        cw.set_robot_model(rmodel, sc.default_params(rmodel), 0.0, np.zeros((W, n), np.float32))
        for _ in range(3):
            cw.imitation_block(0.0125, 20, graph=fused)
        res.append((cw.get_states(), cw.get_robot(), cw.get_goals()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])
    assert np.max(np.abs(res[0][1][:, 0:2] - robot[:, 0:2])) > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("hmodel,rmodel,n", [("hsfm_farina", "sfm_helbing", 25), ("hsfm_farina", "hsfm_new_guo", 25), ("sfm_guo", "hsfm_farina", 10),
                                             ("hsfm_new_moussaid", "sfm_moussaid", 7), ("sfm_helbing", "hsfm_guo", 40)])
def test_imitation_block_with_a_visible_robot_is_one_launch_and_equals_the_alternating_launches(hmodel, rmodel, n):
    """cs_imitation_block with a VISIBLE robot (robot and crowd act on each other in every substep): the robot's own motion model runs
    inside the crowd's fused launch as the last row of every world (k_sfm_step<..., LEAN = 4>) == 20 x { cs_robot_model_step ;
    cs_step(1) }, bit for bit -- robot rows, crowd rows, goal lists, the robot model's remembered desired force -- over three blocks
    (hybrid worlds: goal switches and respawns that take the robot's position into account)."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W = 37
    S, goals, P, rb = sc.hybrid_worlds(W, n, hmodel, seed0=11)
    rng = np.random.default_rng(n)
    robot = np.zeros((W, 13), np.float32)
    robot[:, 0:2] = rng.uniform(-3, 3, (W, 2)); robot[:, 2] = rng.uniform(-3, 3, W); robot[:, 8] = 0.3; robot[:, 9] = 80
    robot[:, 10:12] = -robot[:, 0:2]; robot[:, 12] = 1.0
    S = np.concatenate([S, robot[:, None, :]], axis=1)
    res = []
    for fused in (True, False):
        cw = CrowdWorlds(S, goals, P, None, None, type=hmodel, all_params_equal=True, respawn_bounds=rb,
                         respawn_worlds=(np.arange(W) % 2 == 1).astype(np.int32), robot_row=True, robot=robot)
        cw.set_robot_model(rmodel, sc.default_params(rmodel), 0.01 + 0.05, np.full((W, n + 1), 0.06, np.float32))
        for _ in range(3):
            cw.imitation_block(0.0125, 20, graph=fused)
        res.append((cw.get_states(), cw.get_robot(), cw.get_goals(), cw.d_robot_memory.download()))
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res[0][0][:, n, 0:8], res[0][1][:, 0:8])          # the robot row of the state IS the robot
    assert np.max(np.abs(res[0][1][:, 0:2] - robot[:, 0:2])) > 0.1


@pytest.mark.gpu
def test_first_ever_imitation_block_captured_into_a_graph():
    """Library scratch and stream capture (include/crowdstep.h, cs_reserve_scratch): the very first cs_imitation_block of a
    process-new (stream, size) may not allocate inside a capture -- it is refused with CS_ERR_ARG and the capture stays usable;
    after cs_reserve_scratch the same first-ever call is captured, and replaying the graph equals the eager calls bit for bit.
    Two batches on two streams keep separate scratch blocks: interleaved replays do not disturb each other."""
    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 53, 19                      # a (W, n, n_substeps) no other test uses: the scratch really is new
    S, goals, P, rb = sc.hybrid_worlds(W, n, "hsfm_farina", seed0=9)
    rng = np.random.default_rng(3)
    robot = np.zeros((W, 13), np.float32)
    robot[:, 0:2] = rng.uniform(-3, 3, (W, 2)); robot[:, 8] = 0.3; robot[:, 9] = 80; robot[:, 10:12] = -robot[:, 0:2]; robot[:, 12] = 1.0

    def make(stream):
        cw = CrowdWorlds(S, goals, P, None, None, type="hsfm_farina", all_params_equal=True, respawn_bounds=rb,
                         respawn_worlds=(np.arange(W) % 2 == 1).astype(np.int32), robot=robot, stream=stream)
        cw.set_robot_model("sfm_guo", sc.default_params("sfm_guo"), 0.0, np.zeros((W, n), np.float32))
        return cw

    eager = make(None)
    for _ in range(4):
        eager.imitation_block(0.0125, 23)
    s1, s2 = _lib.stream_create(), _lib.stream_create()
    a, b = make(s1), make(s2)
    # (1) unreserved: refused inside the capture, nothing allocated, the capture itself survives (it records nothing)
    with pytest.raises(ValueError, match="cs_reserve_scratch"):
        with _lib.Graph.capture(s1):
            a.imitation_block(0.0125, 23)
    # (2) reserved: the first-ever call is captured
    a.reserve_scratch(23)
    b.reserve_scratch(23)
    with _lib.Graph.capture(s1) as ga:
        a.imitation_block(0.0125, 23)
    with _lib.Graph.capture(s2) as gb:
        b.imitation_block(0.0125, 23)
    for _ in range(4):                  # interleaved replays on two streams: separate scratch blocks
        ga.launch(); gb.launch()
    _lib.stream_sync(s1); _lib.stream_sync(s2)
    for cw in (a, b):
        np.testing.assert_array_equal(cw.get_states(), eager.get_states())
        np.testing.assert_array_equal(cw.get_robot(), eager.get_robot())
    # (3) a larger block on the same stream after the capture: the pinned block is retired, not freed -- the old graph still replays
    a.imitation_block(0.0125, 40)
    ga.launch()
    _lib.stream_sync(s1)
    assert np.all(np.isfinite(a.get_states()[..., :8]))
