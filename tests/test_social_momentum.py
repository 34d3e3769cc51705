"""SURVEY.md §8 row f4 (remaining crowd models): the social-momentum model.  The numpy oracle against golden G10
captured from the reference (MotionModelManager("social_momentum").update_humans), and the HIP kernel
(cs_step with type CS_SOCIAL_MOMENTUM) against both."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from golden_io import load_cases  # noqa: E402

from oracle import crowd_oracle as orc  # noqa: E402


def test_oracle_matches_golden_g10():
    n_cases = 0
    for c in load_cases("g10_social_momentum"):
        rb = c["robot"] if c["robot_visible"] else None
        p, v, g, rew, ch = orc.social_momentum_step(c["in_pos"], c["in_vel"], c["radius"], c["safety"], c["vd"], c["in_goals"],
                                                    c["dt"], rb)
        np.testing.assert_allclose(p, c["out_pos"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(v, c["out_vel"], rtol=0, atol=1e-13)
        np.testing.assert_array_equal(g, c["out_goals"])
        n_cases += 1
    assert n_cases >= 90


def _check_choices(c, w, rb, got_vel, tol=2e-5):
    """The kernel's arg-max is float32 and the model tests the sign of a product that is exactly zero in exact
    arithmetic for parallel-moving pairs (oracle docstring).  Every chosen velocity must be a collision-free action; it
    must be optimal for SOME resolution of those sign coin-flips: the best reward it can get is not below the smallest
    reward the best competitor is sure of.  Returns (#humans choosing differently from the reference, #of those with no
    ambiguous pair at all = plain float32 near-ties)."""
    _, v, _, rew, ch, (lo, hi) = orc.social_momentum_step(c["in_pos"], c["in_vel"], c["radius"], c["safety"], c["vd"],
                                                           c["in_goals"], c["dt"], rb, amb_eps=1e-5)
    A = rew.shape[1]
    ang = (2 * np.pi / A) * np.arange(A)
    n_diff = n_plain = 0
    for i in range(c["n"]):
        if np.allclose(got_vel[i], v[i], atol=2e-6):
            continue
        acts = np.stack([np.cos(ang), np.sin(ang)], -1) * c["vd"][i]
        k = int(np.argmin(np.linalg.norm(acts - got_vel[i], axis=-1)))
        if ch[i] < 0:
            assert np.all(got_vel[i] == 0), (w, i, "no action is collision-free: the velocity must be zero")
            continue
        assert np.allclose(acts[k], got_vel[i], atol=2e-6), (w, i, got_vel[i])
        assert np.isfinite(rew[i, k]), (w, i, k, "the chosen action collides")
        need = np.max(lo[i])
        assert hi[i, k] >= need - tol * abs(need), (w, i, k, hi[i, k], need)
        n_diff += 1
        n_plain += bool(np.all(lo[i] == hi[i]))
    return n_diff, n_plain


@pytest.mark.gpu
def test_kernel_matches_golden_g10():
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    humans = differ = plain = 0
    for w, c in enumerate(load_cases("g10_social_momentum")):
        n = c["n"]
        rows = n + int(c["robot_visible"])
        S = np.zeros((1, rows, 13), np.float32)
        S[0, :n, 0:2] = c["in_pos"]; S[0, :n, 3:5] = c["in_vel"]; S[0, :n, 8] = c["radius"]; S[0, :n, 9] = 75
        S[0, :n, 10:12] = c["in_goals"][:, 0]; S[0, :n, 12] = c["vd"]
        safety = np.zeros((1, rows), np.float32)
        safety[0, :n] = c["safety"]
        robot = None
        rb = c["robot"] if c["robot_visible"] else None
        if rb is not None:
            robot = np.zeros((1, 13), np.float32)
            robot[0, 0:2] = rb[0:2]; robot[0, 3:5] = rb[2:4]; robot[0, 8] = rb[4]
            S[0, n] = robot[0]
            safety[0, n] = rb[5]
        cw = CrowdWorlds(S, c["in_goals"][None], None, safety, None, type="social_momentum", robot_row=bool(c["robot_visible"]),
                         robot=robot)
        peek = cw.peek(c["dt"])[0]
        cw.step(c["dt"], 1)
        got = cw.get_states()[0]
        np.testing.assert_allclose(got[:n, 0:2], c["out_pos"], rtol=0, atol=2e-6)       # p += v dt: no decision involved
        np.testing.assert_array_equal(cw.get_goals()[0], c["out_goals"].astype(np.float32))
        np.testing.assert_array_equal(got[:n, 10:12], c["out_goals"][:, 0].astype(np.float32))
        nd, npl = _check_choices(c, w, rb, got[:n, 3:5])
        differ += nd; plain += npl
        np.testing.assert_array_equal(peek[:, [0, 1, 3, 4]], got[:n][:, [0, 1, 3, 4]])   # peek = the same step, uncommitted
        if rb is not None:
            np.testing.assert_array_equal(got[n, 0:2], robot[0, 0:2])                    # nobody moves the robot here
        humans += n
    print(f"social momentum: {humans} humans, {differ} chose another action than the float64 reference "
          f"(all justified by a sign coin-flip of the reference), {plain} of them plain float32 near-ties")
    assert humans > 900 and plain <= humans // 100 and differ <= humans // 3


@pytest.mark.gpu
def test_batched_social_momentum_blocks_and_respawn():
    """Fused substeps of many worlds = the same worlds stepped one by one; parallel-traffic respawn keeps everybody inside."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 300, 12
    pos, yaw, g = sc.parallel_traffic(W, n, seed0=4000)
    S = sc.make_states(pos, yaw, g)
    S[:, :, 3] = -1.0                                    # walking towards x = -L/2 - 3
    cw = CrowdWorlds(S, g, None, None, None, type="social_momentum", respawn_bounds=(7.0, 1.5))
    one = CrowdWorlds(S[:7], g[:7], None, None, None, type="social_momentum", respawn_bounds=(7.0, 1.5))
    for _ in range(12):
        cw.step(0.25, 4)
        for _ in range(4):
            one.step(0.25, 1)
    A, B = cw.get_states(), one.get_states()
    np.testing.assert_array_equal(A[:7], B)
    assert np.all(np.isfinite(A[..., :5])) and np.all(A[..., 0] > -10.5) and np.all(np.abs(A[..., 1]) <= 1.5 + 1e-6)
    sp = np.linalg.norm(A[..., 3:5], axis=-1)
    assert np.all((np.abs(sp - 1.0) < 1e-6) | (sp == 0))  # every velocity is one of the 20 actions (or none was free)
    assert np.any(A[..., 0] > 7.0)                         # somebody was respawned behind the others


@pytest.mark.gpu
def test_motion_model_manager_facade_social_momentum():
    """MotionModelManager("social_momentum").update_humans on agent objects, as a reference caller would use it."""
    from social_navigation_pyenvs_amd.social_gym.src.agent import HumanAgent, RobotAgent
    from social_navigation_pyenvs_amd.social_gym.src.motion_model_manager import MotionModelManager

    checked = 0
    for c in load_cases("g10_social_momentum")[::7]:
        n = c["n"]
        humans = []
        for i in range(n):
            goals = [[float(g[0]), float(g[1])] for g in c["in_goals"][i] if not np.isnan(g[0])]
            h = HumanAgent(None, i, "sfm_helbing", [float(c["in_pos"][i, 0]), float(c["in_pos"][i, 1])], 0.0, goals,
                           radius=float(c["radius"][i]), mass=75, des_speed=float(c["vd"][i]))
            h.linear_velocity = c["in_vel"][i].copy()
            h.safety_space = float(c["safety"][i])
            humans.append(h)
        robot = RobotAgent(None)
        if c["robot_visible"]:
            rb = c["robot"]
            robot = RobotAgent(None, pos=(rb[0], rb[1]), radius=float(rb[4]), goals=[[4.0, 4.0]])
            robot.linear_velocity = rb[2:4].copy()
            robot.safety_space = float(rb[5])
        mm = MotionModelManager("social_momentum", bool(c["robot_visible"]), False, humans, robot, [])
        assert len(humans[0].action_set) == 20
        mm.update_humans(0.0, c["dt"])
        got_p = np.array([h.position for h in humans])
        np.testing.assert_allclose(got_p, c["out_pos"], rtol=0, atol=2e-6)
        for i, h in enumerate(humans):
            want = [list(g) for g in c["out_goals"][i] if not np.isnan(g[0])]
            np.testing.assert_allclose(np.array(h.goals), np.array(want), atol=1e-6)
            sp = np.linalg.norm(h.linear_velocity)
            assert sp == 0 or abs(sp - c["vd"][i]) < 1e-6
        checked += 1
    assert checked >= 10
