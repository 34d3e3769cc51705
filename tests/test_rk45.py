"""RK45 integration of the crowd (MotionModelManager(runge_kutta=True).update_humans).  Golden G12 was
recorded from the reference (tests/golden/make_golden.py gen_g12_rk45): rows before / after every call and the number of
right-hand-side evaluations, which pins scipy's step-size control."""
import numpy as np
import pytest

from golden_io import load_cases
from oracle import crowd_oracle as orc


def test_oracle_rk45_matches_reference_g12():
    """scipy's solve_ivp around the restated right-hand side, call by call from the recorded state."""
    worst = 0.0
    for ci, c in enumerate(load_cases("g12_rk45")):
        for k in range(0, len(c["nfev"]), 3):
            sim = orc.Rk45Crowd(c["rows"][k], c["goals"][k], c["params"], c["model"], c["all_params_equal"], c["walls"], c["robot"])
            nf = sim.update_humans(0.0, c["dt"])
            assert nf == c["nfev"][k], (ci, k, nf, c["nfev"][k])
            err = np.max(np.abs(sim.rows - c["rows"][k + 1]))
            worst = max(worst, err)
            got_goals = np.full_like(c["goals"][k + 1], np.nan)
            for i, gl in enumerate(sim.goals):
                got_goals[i, :len(gl)] = gl
            np.testing.assert_array_equal(got_goals, c["goals"][k + 1])
    assert worst < 1e-8, worst


def _worlds(c, k):
    rows, goals = c["rows"][k], c["goals"][k]
    n = c["n"]
    S = np.zeros((n + int(c["robot_visible"]), 13), np.float32)
    S[:n] = rows[:, :13]
    safety = np.zeros(len(S), np.float32)
    safety[:n] = rows[:, 13]
    if c["robot_visible"]:
        rb = c["robot"]
        S[n, 0:2], S[n, 3:5], S[n, 8], S[n, 9], S[n, 12] = rb[0:2], rb[2:4], rb[4], 80.0, 1.0
        safety[n] = rb[5]
    return S, goals, safety


def _run_case_call(c, k):
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    S, goals, safety = _worlds(c, k)
    walls = c["walls"] if c["walls"].shape[0] else None
    cw = CrowdWorlds(S, goals, c["params"], safety, walls, type=c["model"], all_params_equal=c["all_params_equal"],
                     robot_row=c["robot_visible"])
    nfev = cw.update_humans_rk45(c["dt"], desired_force=c["rows"][k][:, 14:16])
    return cw, int(nfev[0])


@pytest.mark.gpu
def test_rk45_kernel_matches_reference_g12_substep_sized_calls():
    """dt = 0.0125 (the simulator's sampling time): cs_update_humans_rk45 call by call from the recorded state, float32 kernel
    vs the float64 reference -- the same number of right-hand-side evaluations (= the same accept / reject decisions and step
    sizes, including calls with rejected steps) and the same rows to 1e-5, in at least 99 % of the calls."""
    calls = other = 0
    for ci, c in enumerate(load_cases("g12_rk45")):
        if c["dt"] > 0.1:
            continue
        n = c["n"]
        for k in range(len(c["nfev"])):
            cw, nfev = _run_case_call(c, k)
            got, ref = cw.get_states()[0][:n], c["rows"][k + 1]
            err = np.max(np.abs(got[:, [0, 1, 3, 4]] - ref[:, [0, 1, 3, 4]]))
            calls += 1
            if nfev != c["nfev"][k]:   # an error estimate on the accept / reject edge (stiff wall contact): another step sequence
                other += 1
                assert err < 5e-3 and c["nfev"][k] > 8, (ci, c["model"], k, nfev, c["nfev"][k], err)
                continue
            assert err < 1e-5, (ci, c["model"], n, k, err)
            if c["model"].startswith("hsfm"):
                dth = np.abs((got[:, 2] - ref[:, 2] + np.pi) % (2 * np.pi) - np.pi)
                assert np.max(dth) < 1e-4 and np.max(np.abs(got[:, 5:7] - ref[:, 5:7])) < 1e-4, (ci, k)
                assert np.max(np.abs(got[:, 7] - ref[:, 7])) < 2e-3 * max(1.0, np.max(np.abs(ref[:, 7]))), (ci, k)
            np.testing.assert_allclose(cw.get_goals()[0], c["goals"][k + 1], atol=1e-6)
            np.testing.assert_allclose(got[:, 10:12], ref[:, 10:12], atol=1e-6)
    assert calls >= 600 and other <= 0.01 * calls, (calls, other)


@pytest.mark.gpu
def test_rk45_kernel_matches_reference_g12_quarter_second_calls():
    """dt = 0.25: 2 to 80 adaptive steps per call, some through stiff contacts (k1 = 1.2e5 N/m) where float32 rounding is
    amplified inside one call and an error estimate near 1 flips an accept / reject decision.  Asserted: the step sequence
    agrees in >= 80 % of the calls, those agree to 1e-4 in >= 70 % of them, and every call stays within what the solver's
    own tolerance (rtol 1e-3) allows for a different step sequence."""
    total = same = tight = 0
    for ci, c in enumerate(load_cases("g12_rk45")):
        if c["dt"] < 0.1:
            continue
        n = c["n"]
        for k in range(len(c["nfev"])):
            cw, nfev = _run_case_call(c, k)
            got, ref = cw.get_states()[0][:n], c["rows"][k + 1]
            err = np.max(np.abs(got[:, [0, 1]] - ref[:, [0, 1]]))
            total += 1
            same += int(nfev == c["nfev"][k])
            tight += int(nfev == c["nfev"][k] and err < 1e-4)
            assert err < 5e-2, (ci, c["model"], k, err, nfev, c["nfev"][k])
            # (a stiff contact inside the quarter second can double the number of steps once the first accept / reject decision differs)
            assert abs(nfev - c["nfev"][k]) <= max(12, 1.5 * c["nfev"][k]), (ci, k, nfev, c["nfev"][k])
    assert same >= 0.8 * total and tight >= 0.7 * same, (total, same, tight)


@pytest.mark.gpu
def test_rk45_batch_of_worlds_and_layouts():
    """W different worlds in one launch == one launch per world; SoA == AoS."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    cases = [c for c in load_cases("g12_rk45") if c["n"] == 9 and c["model"] == "hsfm_farina"]
    c = cases[0]
    ks = list(range(6))
    Ss, Gs, Fs = zip(*[_worlds(c, k) for k in ks])
    walls = c["walls"] if c["walls"].shape[0] else None
    mem = np.stack([c["rows"][k][:, 14:16] for k in ks])
    outs = {}
    for layout in ("aos", "soa"):
        cw = CrowdWorlds(np.stack(Ss), np.stack(Gs), c["params"], np.stack(Fs), walls, type=c["model"],
                         all_params_equal=c["all_params_equal"], robot_row=c["robot_visible"], layout=layout)
        nf = cw.update_humans_rk45(c["dt"], desired_force=mem)
        outs[layout] = (cw.get_states(), nf)
    np.testing.assert_array_equal(outs["aos"][0], outs["soa"][0])
    np.testing.assert_array_equal(outs["aos"][1], outs["soa"][1])
    cw = CrowdWorlds(np.stack(Ss), np.stack(Gs), np.tile(c["params"], (len(ks), 1, 1)), np.stack(Fs), walls, type=c["model"],
                     all_params_equal=c["all_params_equal"], robot_row=c["robot_visible"])   # per-world parameter blocks
    nf = cw.update_humans_rk45(c["dt"], desired_force=mem)
    np.testing.assert_array_equal(cw.get_states(), outs["aos"][0])
    np.testing.assert_array_equal(nf, outs["aos"][1])
    for i, k in enumerate(ks):
        cw = CrowdWorlds(Ss[i], Gs[i], c["params"], Fs[i], walls, type=c["model"], all_params_equal=c["all_params_equal"],
                         robot_row=c["robot_visible"])
        nf = cw.update_humans_rk45(c["dt"], desired_force=mem[i])
        np.testing.assert_array_equal(cw.get_states()[0], outs["aos"][0][i])
        assert nf[0] == outs["aos"][1][i]


@pytest.mark.gpu
def test_motion_model_manager_runge_kutta_facade_g12():
    """MotionModelManager(..., runge_kutta=True).update_humans(t, dt) on HumanAgent objects rebuilt from the recorded rows
    (scenes without walls), call by call."""
    from social_navigation_pyenvs_amd.social_gym.src.agent import HumanAgent, RobotAgent
    from social_navigation_pyenvs_amd.social_gym.src.motion_model_manager import MotionModelManager

    done = 0
    for ci, c in enumerate(load_cases("g12_rk45")):
        if c["walls"].shape[0] or c["dt"] > 0.1:
            continue
        n = c["n"]
        for k in (0, 7, 19):
            rows, goals = c["rows"][k], c["goals"][k]
            humans = []
            for i in range(n):
                gl = [list(g) for g in goals[i] if not np.any(np.isnan(g))]
                h = HumanAgent(None, i, c["model"], list(rows[i, 0:2]), float(rows[i, 2]), gl, radius=float(rows[i, 8]),
                               mass=float(rows[i, 9]), des_speed=float(rows[i, 12]))
                h.linear_velocity[:] = rows[i, 3:5]
                h.body_velocity[:] = rows[i, 5:7]
                h.angular_velocity = float(rows[i, 7])
                for name, val in zip(("relaxation_time", "Ai", "Aw", "Bi", "Bw", "Ci", "Cw", "Di", "Dw", "Ei", "k1", "k2", "agent_lambda",
                                      "gamma", "ns", "ns1", "ko", "kd", "alpha", "k_lambda"), c["params"][i]):
                    if hasattr(h, name):
                        setattr(h, name, float(val))
                humans.append(h)
            robot = RobotAgent(None)
            if c["robot_visible"]:
                rb = c["robot"]
                robot.position[:] = rb[0:2]
                robot.linear_velocity[:] = rb[2:4]
                robot.radius = float(rb[4])
                robot.goals = [[4.0, 4.0]]
            mm = MotionModelManager(c["model"], c["robot_visible"], True, humans, robot, [])
            assert mm.all_equal_humans == c["all_params_equal"]
            if rows[0, 13] > 0:
                mm.set_safety_space(float(rows[0, 13]) - 0.01)
            mm._desired_force = rows[:, 14:16].copy()
            mm.update_humans(0.0, c["dt"])
            ref = c["rows"][k + 1]
            if mm.rk45_nfev != c["nfev"][k]:
                continue
            got = np.array([[*h.position, h.yaw, *h.linear_velocity, *h.body_velocity] for h in humans])
            assert np.max(np.abs(got[:, [0, 1, 3, 4]] - ref[:, [0, 1, 3, 4]])) < 1e-5, (ci, k)
            assert [list(map(float, g)) for g in humans[0].goals] == [list(g) for g in c["goals"][k + 1][0] if not np.any(np.isnan(g))]
            done += 1
    assert done >= 20


@pytest.mark.gpu
@pytest.mark.parametrize("n,model,robot", [(1, "sfm_helbing", False), (1, "hsfm_new_guo", True), (2, "sfm_moussaid", False),
                                           (63, "hsfm_farina", True), (64, "sfm_guo", False)])
def test_rk45_edge_sizes_against_the_oracle(n, model, robot):
    """One human, and worlds that fill the wavefront (64 rows): kernel vs the scipy-based oracle from float32-rounded inputs."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rng = np.random.default_rng(n * 7 + int(robot))
    side = int(np.ceil(np.sqrt(n + 1)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    pts = (np.stack([gx.ravel(), gy.ravel()], -1)[:n + 1] - side / 2) * 1.1 + rng.uniform(-0.15, 0.15, (n + 1, 2))
    rows = np.zeros((n, 16))
    rows[:, 0:2] = pts[:n]
    rows[:, 2] = rng.uniform(-np.pi, np.pi, n)
    rows[:, 5:7] = rng.normal(0, 0.3, (n, 2))
    c, s = np.cos(rows[:, 2]), np.sin(rows[:, 2])
    if model.startswith("hsfm"):
        rows[:, 3] = c * rows[:, 5] - s * rows[:, 6]; rows[:, 4] = s * rows[:, 5] + c * rows[:, 6]
    else:
        rows[:, 3:5] = rng.normal(0, 0.3, (n, 2)); rows[:, 5:7] = 0
    rows[:, 7] = rng.normal(0, 0.2, n) if model.startswith("hsfm") else 0.0
    rows[:, 8], rows[:, 9], rows[:, 12], rows[:, 13] = 0.3, 75.0, 1.0, 0.01
    goals = np.stack([-rows[:, 0:2] * 2 + 0.3, rows[:, 0:2]], 1)
    rows[:, 10:12] = goals[:, 0]
    rows = rows.astype(np.float32).astype(np.float64)
    goals = goals.astype(np.float32).astype(np.float64)
    P = np.tile(sc.default_params(model), (n, 1)).astype(np.float32)
    rb = np.array([*pts[n], 0.2, -0.1, 0.3, 0.0], dtype=np.float32).astype(np.float64) if robot else np.zeros(0)
    sim = orc.Rk45Crowd(rows, goals, P.astype(np.float64), model, True, None, rb)
    nf_ref = sim.update_humans(0.0, 0.0125)
    S = np.zeros((n + int(robot), 13), np.float32)
    S[:n] = rows[:, :13]
    safety = np.full(n + int(robot), 0.01, np.float32)
    if robot:
        S[n, 0:2], S[n, 3:5], S[n, 8], S[n, 9], S[n, 12] = rb[0:2], rb[2:4], rb[4], 80.0, 1.0
        safety[n] = 0.0
    cw = CrowdWorlds(S, goals, P, safety, None, type=model, all_params_equal=True, robot_row=robot)
    nf = cw.update_humans_rk45(0.0125)
    assert nf[0] == nf_ref
    got = cw.get_states()[0][:n]
    tol = 2e-3 if model.endswith("moussaid") else 1e-5
    assert np.max(np.abs(got[:, [0, 1, 3, 4]] - sim.rows[:, [0, 1, 3, 4]])) < tol
    with pytest.raises(ValueError, match="64 rows"):
        CrowdWorlds(np.zeros((1, 65, 13), np.float32), np.zeros((1, 65, 1, 2), np.float32), np.zeros((65, 20), np.float32), None, None,
                    type=model).update_humans_rk45(0.0125)


# ------------------------------------------------------------------------------------------------------------------------------
# The rest of the RK45 surface (golden G14, tests/golden/make_golden.py gen_g14_rk45_more): complete_rk45_simulation (dense
# output), update_humans + the respawn rule, the robot under RK45.
# ------------------------------------------------------------------------------------------------------------------------------
def _g14(family):
    return [c for c in load_cases("g14_rk45_more") if c["family"] == family]


def test_oracle_rk45_more_matches_reference_g14():
    """The oracle's restatements against what the reference returned: one solve with t_eval (scipy's dense output on both sides),
    the respawn rule on the agent objects behind a solve, the robot's own solve -- identical numbers of right-hand-side evaluations."""
    for c in _g14("complete"):
        sim = orc.Rk45Crowd(c["rows0"], c["goals0"], c["params"], c["model"], c["all_params_equal"], c["walls"], None)
        hs = sim.complete_simulation(0.0, c["dt"], c["final_time"])
        assert sim.nfev == c["nfev"] and hs.shape == c["human_states"].shape
        assert np.max(np.abs(hs - c["human_states"])) < 1e-6 and np.max(np.abs(sim.rows[:, :13] - c["rows1"][:, :13])) < 1e-6
    moved = 0
    for c in _g14("respawn"):
        for k in range(len(c["nfev"])):
            sim = orc.Rk45Crowd(c["rows"][k], c["goals"][k], c["params"], c["model"], c["all_params_equal"], None, c["robot"] if c["robot_visible"] else None)
            assert sim.update_humans(0.0, c["dt"]) == c["nfev"][k]
            sim.respawn(*c["respawn_bounds"])
            assert np.max(np.abs(sim.rows[:, :13] - c["rows"][k + 1][:, :13])) < 1e-9
            moved += int(np.sum(np.abs(c["rows"][k + 1][:, 0] - c["rows"][k][:, 0]) > 3))
    assert moved >= 8
    for c in _g14("robot"):
        S, n = c["mm_states"], c["n"]
        for k in range(len(c["nfev"])):
            sim = orc.Rk45Robot(c["robots"][k], c["robot_params"], c["robot_model"], S[:n, 0:2], S[:n, 3:5], S[:n, 8], c["human_safety"],
                                c["walls"] if c["walls"].shape[0] else None)
            assert sim.update(0.0, c["dt"]) == c["nfev"][k]
            assert np.max(np.abs(sim.row[:8] - c["robots"][k + 1][:8])) < 1e-8


def _crowd_from_rows(rows, goals, c, robot_row=False, respawn_bounds=None):
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    n = rows.shape[0]
    S = np.zeros((n + int(robot_row), 13), np.float32)
    S[:n] = rows[:, :13]
    safety = np.zeros(len(S), np.float32)
    safety[:n] = rows[:, 13]
    if robot_row:
        rb = c["robot"]
        S[n, 0:2], S[n, 3:5], S[n, 8], S[n, 9], S[n, 12] = rb[0:2], rb[2:4], rb[4], 80.0, 1.0
        safety[n] = rb[5]
    walls = c["walls"] if ("walls" in c and c["walls"].shape[0]) else None
    return CrowdWorlds(S, goals, c["params"], safety, walls, type=c["model"], all_params_equal=c["all_params_equal"], robot_row=robot_row,
                       respawn_bounds=respawn_bounds)


@pytest.mark.gpu
def test_complete_rk45_simulation_g14():
    """cs_complete_rk45_simulation (one float32 solve over 1.5 - 2 s, the solution at np.arange(0, final_time, dt) from the dense
    output) against what the reference's complete_rk45_simulation returned: where the kernel takes the same number of right-hand-side
    evaluations (= the same step sequence) the sampled states agree to 2e-4 (float32 over ~20 adaptive steps); a different sequence
    stays inside the solver's own tolerance.  t_eval[0] is the initial state exactly."""
    total = same = tight = 0
    for c in _g14("complete"):
        cw = _crowd_from_rows(c["rows0"], c["goals0"], c)
        n_eval = c["human_states"].shape[0]
        hs, nfev = cw.complete_rk45_simulation(c["dt"], c["final_time"], n_eval, desired_force=c["rows0"][:, 14:16])
        ref = c["human_states"]
        assert hs.shape == (1,) + ref.shape
        np.testing.assert_allclose(hs[0, 0], ref[0].astype(np.float32), atol=1e-6)          # x = 0 of the first step's interpolant
        err = np.max(np.abs(hs[0][..., 0:2] - ref[..., 0:2]))
        if c["model"].startswith("hsfm_new"):
            # the torque-on-total-force models are the reference's unstable ones (|omega| runs away, hundreds of tiny steps): a 2 s
            # solve is chaotic in any arithmetic -- the first samples and the order of magnitude of the work are what can be compared
            assert np.all(np.isfinite(hs)) and np.max(np.abs(hs[0][:2][..., 0:2] - ref[:2][..., 0:2])) < 2e-2, c["model"]
            assert 0.3 * c["nfev"] <= int(nfev[0]) <= 3.0 * c["nfev"], (c["model"], int(nfev[0]), c["nfev"])
            continue
        total += 1
        if int(nfev[0]) == c["nfev"]:
            same += 1
            assert err < 5e-3, (c["model"], c["n"], err)          # (1.5 - 2 s through contacts at 120 kN/m in float32)
            tight += int(err < 2e-4)
            got = cw.get_states()[0]
            assert np.max(np.abs(got[:, [0, 1]] - c["rows1"][:, [0, 1]])) < max(2e-4, 2 * err)    # left at the state of t + final_time
        assert err < 5e-2 and abs(int(nfev[0]) - c["nfev"]) <= max(12, 0.5 * c["nfev"]), (c["model"], err, int(nfev[0]), c["nfev"])
    assert same >= 0.6 * total and tight >= 0.7 * same, (same, tight, total)


@pytest.mark.gpu
def test_rk45_with_the_respawn_rule_g14():
    """update_humans of an RK45 crowd in a parallel-traffic scene (cs_update_humans_rk45 with CS_RESPAWN): the solve, then the respawn
    rule of the agent objects (position and goal list; no column-6:8 write) -- call by call from the recorded state."""
    moved = 0
    for c in _g14("respawn"):
        for k in range(len(c["nfev"])):
            cw = _crowd_from_rows(c["rows"][k], c["goals"][k], c, robot_row=c["robot_visible"], respawn_bounds=c["respawn_bounds"])
            nfev = cw.update_humans_rk45(c["dt"], desired_force=c["rows"][k][:, 14:16])
            ref = c["rows"][k + 1]
            got = cw.get_states()[0][:c["n"]]
            if int(nfev[0]) != c["nfev"][k]:
                continue
            # (three humans were put on the edge of the zone, next to others: contacts; hsfm_new* is the stiffest model)
            assert np.max(np.abs(got[:, [0, 1, 3, 4]] - ref[:, [0, 1, 3, 4]])) < (5e-4 if c["model"].startswith("hsfm_new") else 1e-5), (c["model"], k)
            np.testing.assert_allclose(got[:, 5:7], ref[:, 5:7], atol=2e-4)                    # body velocity untouched by the respawn
            np.testing.assert_allclose(cw.get_goals()[0], c["goals"][k + 1], atol=1e-5)
            moved += int(np.sum(np.abs(got[:, 0] - c["rows"][k][:, 0]) > 3))
    assert moved >= 8


@pytest.mark.gpu
def test_robot_under_rk45_g14():
    """cs_robot_model_rk45 (update_robot of a robot whose SFM / HSFM model was set with runge_kutta=True) call by call from the
    recorded robot row, humans standing: the same number of right-hand-side evaluations and the robot's row to 1e-5 (dt = 0.0125)
    / 1e-4 (dt = 0.25) in nearly every call."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    calls = same = 0
    for c in _g14("robot"):
        S, n = c["mm_states"], c["n"]
        walls = c["walls"] if c["walls"].shape[0] else None
        hm = np.asarray(c["human_safety"], np.float32)
        for k in range(len(c["nfev"])):
            row = c["robots"][k]
            rb = np.zeros(13, np.float32)
            rb[:13] = row[:13]
            cw = CrowdWorlds(S[:n].astype(np.float32), np.zeros((n, 1, 2), np.float32), np.zeros((n, 20), np.float32), None, walls, type="sfm_helbing",
                             robot=rb)
            cw.set_robot_model(c["robot_model"], c["robot_params"], float(row[13]), hm[None])
            cw.d_robot_memory.upload(np.asarray(row[14:16], np.float32).reshape(1, 2))
            nfev = cw.robot_model_rk45(c["dt"])
            got, ref = cw.get_robot()[0], c["robots"][k + 1]
            calls += 1
            if int(nfev[0]) != c["nfev"][k]:
                assert np.max(np.abs(got[[0, 1, 3, 4]] - ref[[0, 1, 3, 4]])) < 5e-3, (c["robot_model"], k)
                continue
            same += 1
            tol = 2e-5 if c["dt"] < 0.1 else (1e-3 if c["robot_model"].startswith("hsfm_new") else 1e-4)
            assert np.max(np.abs(got[[0, 1, 3, 4]] - ref[[0, 1, 3, 4]])) < tol, (c["robot_model"], k, np.max(np.abs(got[:8] - ref[:8])))
            if c["robot_model"].startswith("hsfm"):
                dth = abs((got[2] - ref[2] + np.pi) % (2 * np.pi) - np.pi)
                assert dth < 10 * tol and np.max(np.abs(got[5:7] - ref[5:7])) < 10 * tol
    assert calls >= 200 and same >= 0.95 * calls, (calls, same)


# ---------------------------------------------------------------------------------------------------------------------
# The robot under RK45 at the Gym seam (golden G15, tests/golden/make_golden.py gen_g15_imitation_rk45):
# set_human_motion_model_as_robot_policy(model, runge_kutta=True) + imitation_learning_step (social_nav_sim.py:862-873,
# motion_model_manager.py:631-640, social_nav_gym.py:252-274)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gym_imitation_learning_step_with_the_robot_under_rk45_g15():
    """The single-env facade, asked for BEFORE the first reset (the pending case) and re-synchronised with the reference before
    every Gym step: 20 x { RK45 solve of the robot over dt ; Euler substep of the crowd } against what the reference returned."""
    from test_facade_cpu import make_env

    steps = 0
    for ci, c in enumerate(load_cases("g15_imitation_rk45")):
        env = make_env(c["model"], c["scenario"], c["human_num"], c["robot_visible"], False)
        env.set_human_motion_model_as_robot_policy(c["robot_model"], True)          # no world yet: kept until the first reset
        if c["safety_space"] > 0:
            env.set_safety_space(c["safety_space"])
        env.reset(phase=c["phase"], test_case=c["test_case"])
        mm = env.motion_model_manager
        assert mm.robot_motion_model_title == c["robot_model"] and mm.robot_runge_kutta is True
        np.testing.assert_allclose(mm.states, c["mm_states"][0], atol=1e-12)
        for k in range(len(c["rewards"])):
            mm.states[...] = c["mm_states"][k]
            mm.goals[...] = c["mm_goals"][k]
            mm._sync_goal_lists_from_array()
            r = c["robots"][k]
            env.robot.position, env.robot.yaw, env.robot.linear_velocity = r[0:2].copy(), float(r[2]), r[3:5].copy()
            env.robot.body_velocity, env.robot.angular_velocity = r[5:7].copy(), float(r[7])
            env.robot.desired_force = r[14:16].copy()
            env.global_time = 0.25 * k
            ob, reward, term, trunc, info = env.imitation_learning_step()
            assert mm.robot_rk45_nfev >= 20 * 8                                      # 20 solves, at least one accepted step each
            # 20 float32 substeps of a stiff system vs the float64 reference (the secondary, end-of-step bound of test_imitation.py)
            tol = 3e-4 if c["respawn"] else 5e-5
            ref = c["robots"][k + 1]
            got = np.array([*env.robot.position, env.robot.yaw, *env.robot.linear_velocity, *env.robot.body_velocity, env.robot.angular_velocity])
            assert np.max(np.abs(got[[0, 1, 3, 4]] - ref[[0, 1, 3, 4]])) < tol, (ci, c["robot_model"], c["model"], k, np.abs(got - ref[:8]))
            if c["robot_model"].startswith("hsfm"):
                assert abs((got[2] - ref[2] + np.pi) % (2 * np.pi) - np.pi) < 10 * tol and np.max(np.abs(got[5:7] - ref[5:7])) < 10 * tol
            obs = np.array([[o.px, o.py, o.vx, o.vy] for o in ob])
            assert np.max(np.abs(obs - c["obs"][k + 1][:, :4])) < tol, (ci, k)
            assert (term, trunc) == (bool(c["terminated"][k]), bool(c["truncated"][k]))
            assert type(info[0]).__name__ == c["infos"][k]
            assert abs(reward - c["rewards"][k]) < 10 * tol
            steps += 1
    assert steps >= 100


@pytest.mark.gpu
@pytest.mark.parametrize("hmodel,rmodel,visible,safety", [("hsfm_farina", "sfm_guo", True, 0.0), ("sfm_helbing", "hsfm_new_guo", False, 0.1)])
def test_batched_imitation_with_the_robot_under_rk45_matches_single_env(hmodel, rmodel, visible, safety):
    """BatchedSocialNavGym.set_human_motion_model_as_robot_policy(..., runge_kutta=True) + imitation_learning_step == W single-world
    facades, step by step; an ORCA robot cannot be integrated by RK45 (the reference raises NotImplementedError, :579)."""
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym
    from social_navigation_pyenvs_amd.social_gym.src.info import INFO_BY_CODE
    from test_facade_cpu import make_config, make_env

    W, steps = 5, 5
    cfg = make_config(hmodel, "circle_crossing", 5, visible)
    benv = BatchedSocialNavGym(cfg, W, robot_visible=visible)
    benv.reset(phase="val", first_case=11, safety_space=safety)
    with pytest.raises(NotImplementedError):
        benv.set_human_motion_model_as_robot_policy("orca", True)
    benv.set_human_motion_model_as_robot_policy(rmodel, True, safety_space=safety)
    euler = BatchedSocialNavGym(cfg, W, robot_visible=visible)
    euler.reset(phase="val", first_case=11, safety_space=safety)
    euler.set_human_motion_model_as_robot_policy(rmodel, False, safety_space=safety)
    hist = [benv.imitation_learning_step() for _ in range(steps)]
    for _ in range(steps):
        euler.imitation_learning_step()
    robots = benv.cw.get_robot()
    d_int = np.max(np.abs(robots[:, 0:2] - euler.cw.get_robot()[:, 0:2]))
    assert 1e-7 < d_int < 5e-2, d_int                      # the two integrators really differ, by an integrator's worth
    for w in range(W):
        env = make_env(hmodel, "circle_crossing", 5, visible)
        env.set_human_motion_model_as_robot_policy(rmodel, True)
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
def test_rk45_robot_in_an_orca_crowd_leaves_the_state_row_to_the_crowd():
    """cs_robot_model_rk45 with CS_ROBOT_ROW in an ORCA crowd moves d_robot only: the crowd's simulator sees the moved robot after its own
    doStep (motion_model_manager.py:389) -- the rule of the Euler path (cs_robot_model_step)."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 3, 6
    pos, yaw, g = sc.circular_crossing(W, n, 4.0, 5)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    R = np.zeros((W, 13), np.float32)
    R[:, 0:2] = (0.0, -4.0); R[:, 2] = np.pi / 2; R[:, 8] = 0.3; R[:, 9] = 80; R[:, 10:12] = (0.0, 4.0); R[:, 12] = 1.0
    St = np.concatenate([S, R[:, None, :]], axis=1)
    res = {}
    for crowd in ("orca", "sfm_helbing"):
        cw = CrowdWorlds(St, g, None if crowd == "orca" else np.tile(sc.default_params(crowd), (n, 1)), np.full((W, n + 1), 0.01, np.float32), None,
                         type=crowd, robot_row=True, robot=R)
        cw.set_robot_model("sfm_helbing", sc.default_params("sfm_helbing"), 0.0, np.zeros((W, n + 1), np.float32))
        cw.robot_model_rk45(0.0125)
        res[crowd] = (cw.get_states()[:, n].copy(), cw.get_robot().copy())
    np.testing.assert_array_equal(res["orca"][0], St[:, n])                       # ORCA crowd: the state row is untouched ...
    assert np.all(res["orca"][1][:, 3:5] != R[:, 3:5])                             # ... the robot rows moved
    np.testing.assert_array_equal(res["sfm_helbing"][0][:, 0:8], res["sfm_helbing"][1][:, 0:8])   # SFM crowd: written through
