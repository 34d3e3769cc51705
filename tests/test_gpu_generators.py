"""the device-side scenario generators (cs_generate_worlds) against the golden vectors G6
captured from the reference itself, and against the host generators (exact restatement, bit-exact on G6)."""
import collections
import os
import sys
import types

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from golden_io import load_cases  # noqa: E402

pytestmark = pytest.mark.gpu

F32_TOL = dict(rtol=3e-7, atol=1e-6)  # f64 rows rounded to f32; device libm cos/sin may differ in the last f64 bit


def _blank(W, n, G, robot_row=False, layout="aos"):
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    rows = n + int(robot_row)
    return CrowdWorlds(np.zeros((W, rows, 13)), np.full((W, n, G, 2), np.nan), np.zeros((n, 20)), None, None,
                       type="sfm_helbing", all_params_equal=True, robot_row=robot_row, robot=np.zeros((W, 13)),
                       respawn_worlds=np.zeros(W, np.int32), layout=layout)


def test_device_generators_match_golden_g6():
    from social_navigation_pyenvs_amd.generators import generate_worlds

    groups = collections.defaultdict(list)
    hybrid = None
    for c in load_cases("g6_generators"):
        if c["generator"] == "hybrid_choice":
            hybrid = c
            continue
        groups[(c["generator"], int(c["n"]), bool(c["insert_robot"]), bool(c["randomize_attributes"]))].append(c)
    checked = 0
    for (name, n, insert_robot, rand_attr), cases in groups.items():
        scenario = {"circular_crossing": "circle_crossing", "circular_crossing_fixed": "circle_crossing"}.get(name, name)
        G = 1 if name == "parallel_traffic" else 2
        cw = _blank(len(cases), n, G)
        seeds = [max(int(c["seed"]), 0) for c in cases]
        status, scn = generate_worlds(cw, scenario, seeds, insert_robot=insert_robot, randomize_attributes=rand_attr,
                                      randomize_positions=(name != "circular_crossing_fixed"))
        assert not status.any()
        S, goals, robot, flags = cw.get_states(), cw.get_goals(), cw.get_robot(), cw.d_world_flags.download()
        for k, c in enumerate(cases):
            np.testing.assert_allclose(S[k, :, 0:2], c["pos"].astype(np.float32), **F32_TOL)
            np.testing.assert_allclose(S[k, :, 2], c["yaw"].astype(np.float32), **F32_TOL)
            assert np.all(S[k, :, 3:8] == 0)
            np.testing.assert_array_equal(S[k, :, 8], c["radius"].astype(np.float32))       # draws only: bit-exact
            np.testing.assert_array_equal(S[k, :, 12], c["des_speed"].astype(np.float32))
            assert np.all(S[k, :, 9] == 75)
            g = np.asarray(c["goals"], np.float64)[:, :G]
            np.testing.assert_allclose(goals[k], g.astype(np.float32), **F32_TOL)
            np.testing.assert_allclose(S[k, :, 10:12], g[:, 0].astype(np.float32), **F32_TOL)
            if c["insert_robot"]:
                np.testing.assert_allclose(robot[k, 0:2], np.asarray(c["robot_pos"], np.float32), **F32_TOL)
                np.testing.assert_allclose(robot[k, 10:12], np.asarray(c["robot_goals"], np.float32)[0], **F32_TOL)
            assert flags[k] == (1 if name == "parallel_traffic" else 0)
            checked += 1
    assert checked > 100
    # the hybrid choice: np.random.choice of two = next_uint32 & 1 on the freshly seeded stream
    seeds = np.asarray(hybrid["seeds"], np.uint32)
    cw = _blank(len(seeds), 5, 2)
    _, scn = generate_worlds(cw, "hybrid_scenario", seeds)
    want = np.array([1 if str(ch) == "parallel_traffic" else 0 for ch in hybrid["choice"]], np.int32)
    np.testing.assert_array_equal(scn, want)


def _config(test_sim="hybrid_scenario", human_num=10, policy="hsfm_farina", randomize="false"):
    import configparser

    cfg = configparser.RawConfigParser()
    cfg.read_dict({
        "env": {"time_limit": 50, "time_step": 0.0125, "robot_time_step": 0.25, "val_size": 100, "test_size": 100,
                "randomize_attributes": randomize},
        "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2, "discomfort_penalty_factor": 0.5},
        "sim": {"train_val_sim": test_sim, "test_sim": test_sim, "square_width": 10, "circle_radius": 7, "human_num": human_num,
                "traffic_length": 14, "traffic_height": 3},
        "humans": {"visible": "true", "policy": policy, "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
        "robot": {"visible": "false", "policy": "none", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
    })
    return cfg


@pytest.mark.parametrize("sim,visible,layout_phase", [("hybrid_scenario", False, "test"), ("circle_crossing", True, "val"),
                                                      ("parallel_traffic", False, "train"),
                                                      ("circular_crossing_with_static_obstacles", True, "test")])
def test_batched_gym_device_reset_equals_host_reset(sim, visible, layout_phase):
    """BatchedSocialNavGym.reset(device=True) fills the same rows as W serial host resets of the reference logic."""
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    W = 96
    host = BatchedSocialNavGym(_config(sim), W, robot_visible=visible)
    host.reset(phase=layout_phase, first_case=7)
    dev = BatchedSocialNavGym(_config(sim), W, robot_visible=visible)
    dev.reset(phase=layout_phase, first_case=7, device=True)
    Sh, Sd = host.cw.get_states(), dev.cw.get_states()
    assert Sh.shape == Sd.shape
    np.testing.assert_allclose(Sd, Sh, **F32_TOL)
    np.testing.assert_allclose(dev.cw.get_goals(), host.cw.get_goals(), **F32_TOL)  # NaN padding compares equal
    np.testing.assert_allclose(dev.cw.get_robot(), host.cw.get_robot(), **F32_TOL)
    if host.cw.d_world_flags is not None:
        np.testing.assert_array_equal(dev.cw.d_world_flags.download(), host.cw.d_world_flags.download())
    assert dev.cw.respawn_bounds == host.cw.respawn_bounds
    # and both step identically from there
    a = np.tile(np.array([[0.3, 0.1]], np.float32), (W, 1))
    oh = host.step(a)
    od = dev.step(a)
    np.testing.assert_allclose(od[0], oh[0], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(od[2], oh[2])


def test_masked_regeneration_and_soa_layout():
    from social_navigation_pyenvs_amd.generators import generate_worlds

    W, n = 130, 12
    cw = _blank(W, n, 2, robot_row=True, layout="soa")
    generate_worlds(cw, "circle_crossing", 1000 + np.arange(W))
    S0, g0 = cw.get_states(), cw.get_goals()
    assert np.all(np.isfinite(S0)) and np.all(S0[:, n, 8] == np.float32(0.3))  # robot row filled
    ref = _blank(W, n, 2, robot_row=True, layout="aos")
    generate_worlds(ref, "circle_crossing", 1000 + np.arange(W))
    np.testing.assert_array_equal(S0, ref.get_states())  # same rows through either layout
    mask = (np.arange(W) % 3 == 0)
    status, scn = generate_worlds(cw, "circle_crossing", 5000 + np.arange(W), mask=mask)
    S1 = cw.get_states()
    np.testing.assert_array_equal(S1[~mask], S0[~mask])
    assert np.all(np.any(S1[mask][:, :n, 0:2] != S0[mask][:, :n, 0:2], axis=(1, 2)))
    np.testing.assert_array_equal(cw.get_goals()[~mask], g0[~mask])
    assert np.all(scn[~mask] == -1) and np.all(scn[mask] == 0)


def test_generator_failures_are_reported_not_hung():
    from social_navigation_pyenvs_amd.generators import generate_worlds

    cw = _blank(8, 60, 2)  # 60 humans cannot stand on a circle of radius 2
    with pytest.raises(RuntimeError):
        generate_worlds(cw, "circle_crossing", np.arange(8), circle_radius=2, max_tries=200)
    status, _ = generate_worlds(cw, "circle_crossing", np.arange(8), circle_radius=2, max_tries=200, raise_on_failure=False)
    assert np.all(status == 1)
    cw = _blank(4, 100, 1)  # 100 x pi 0.3^2 = 28.3 m^2 > 0.4 x 14 x 3
    with pytest.raises(ValueError):
        generate_worlds(cw, "parallel_traffic", np.arange(4))


def test_full_size_device_reset_properties():
    """Full-size resets in one launch each: one GPU's shard of BASELINE.json configs[4] (8192 worlds x 50 humans, circular
    crossing on R = 20: the generator also keeps every human off the earlier humans' goals, 100 blocked spots) and 8192 worlds of the reference's static-obstacle scenario (10 humans: with more, its slots
    i and i + n/2 coincide and the reference's own loop stops terminating).  Placement invariants of the generators
    hold in every world, sampled worlds equal the host generator, and a seed gives the same world in any batch."""
    import time

    from social_navigation_pyenvs_amd.generators import generate_worlds
    from social_navigation_pyenvs_amd.social_gym.social_nav_sim import SocialNavSim

    host = types.SimpleNamespace(_attributes=SocialNavSim._attributes)
    W = 8192
    seeds = 1000 + np.arange(W)
    for scenario, n, R in (("circle_crossing", 50, 20), ("circular_crossing_with_static_obstacles", 10, 7)):
        cw = _blank(W, n, 2)
        t0 = time.perf_counter()
        status, scn = generate_worlds(cw, scenario, seeds, circle_radius=R)
        el = time.perf_counter() - t0
        assert not status.any()
        S, goals = cw.get_states(), cw.get_goals()
        assert np.all(np.isfinite(S)) and np.all(np.isfinite(goals))
        p, r = S[:, :, 0:2].astype(np.float64), S[:, :, 8].astype(np.float64)
        gap = r[:, :, None] + r[:, None, :] + 0.2 - 1e-5
        d = np.linalg.norm(p[:, :, None] - p[:, None, :], axis=-1) + np.eye(n)[None] * 1e9
        assert np.all(d >= gap)                                                  # accepted placements keep the gap
        static = scenario != "circle_crossing"
        if static:
            assert np.all((r[:, :3] > 0.6 - 1e-6) & (r[:, :3] <= 1.0)) and np.all(r[:, 3:] == np.float32(0.3))
            assert np.all(S[:, :3, 12] == 0) and np.all(S[:, 3:, 12] == 1)
            np.testing.assert_array_equal(goals[:, :3, 0], S[:, :3, 0:2])        # obstacles: goal = own position
            np.testing.assert_array_equal(goals[:, 3:, 0], -S[:, 3:, 0:2])
            np.testing.assert_array_equal(goals[:, :, 1], S[:, :, 0:2])
        else:
            assert np.all(r == np.float32(0.3)) and np.all(S[..., 12] == 1)
            np.testing.assert_array_equal(goals[:, :, 0], -S[:, :, 0:2])
            np.testing.assert_array_equal(goals[:, :, 1], S[:, :, 0:2])
            dg = np.linalg.norm(p[:, :, None] + p[:, None, :], axis=-1)           # a later human also keeps off earlier goals
            iu = np.triu_indices(n, 1)
            assert np.all(dg[:, iu[1], iu[0]] >= gap[:, iu[1], iu[0]])
        for w in (0, 4097, W - 1):
            np.random.seed(int(seeds[w]))
            if static:
                data = SocialNavSim.generate_circular_crossing_with_static_obstacles(
                    host, insert_robot=True, human_policy="hsfm_farina", robot_radius=0.3, circle_radius=R, n_actors=n)
            else:
                data = SocialNavSim.generate_circular_crossing_setting(
                    host, insert_robot=True, human_policy="hsfm_farina", robot_radius=0.3, circle_radius=R, n_actors=n,
                    randomize_human_positions=True, randomize_human_attributes=False)
            hp = np.array([data["humans"][i]["pos"] for i in range(n)])
            np.testing.assert_allclose(S[w, :, 0:2], hp.astype(np.float32), **F32_TOL)
        sub = _blank(5, n, 2)
        pick = np.array([17, 4097, 3, 8191, 100])
        generate_worlds(sub, scenario, seeds[pick], circle_radius=R)
        np.testing.assert_array_equal(sub.get_states(), S[pick])                 # batch composition does not matter
        print(f"device reset of {W} x {n} ({scenario}): {el * 1e3:.1f} ms including the seed upload and status download")


def test_device_resident_step_equals_host_step_and_auto_resets():
    """BatchedSocialNavGym.step_device (torch tensors in HBM, no host copies, masked device reset of finished worlds)
    against the host-array step of the same batch, and the regenerated worlds against the host generator."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    W = 64
    cfg = _config("circle_crossing", human_num=6)
    host = BatchedSocialNavGym(cfg, W)
    host.reset(phase="test", first_case=3, device=True)
    dev = BatchedSocialNavGym(cfg, W)
    dev.reset(phase="test", first_case=3, device=True)
    rng = np.random.default_rng(0)
    for k in range(6):   # no episode ends this early: the two paths must agree exactly
        a = rng.uniform(-0.5, 0.5, (W, 2)).astype(np.float32)
        oh, rh, th, uh, ih = host.step(a)
        od, rd, td, ud, idv = dev.step_device(torch.as_tensor(a, device="cuda"))
        assert od.is_cuda and od.shape == (W, 6, 5)
        np.testing.assert_array_equal(od.cpu().numpy(), oh)
        np.testing.assert_array_equal(rd.cpu().numpy(), rh)
        np.testing.assert_array_equal(td.cpu().numpy(), th)
        np.testing.assert_array_equal(idv.cpu().numpy(), ih)
    np.testing.assert_array_equal(dev._dl["gtime"].cpu().numpy(), host.global_time)
    # drive every robot straight to its goal (0, R): ReachGoal ends the episode and the world is regenerated
    seeds0 = dev._dl["seeds"].cpu().numpy().copy()
    ended = np.zeros(W, bool)
    for k in range(80):
        rb = dev.cw.d_robot.torch()
        to_goal = rb[:, 10:12] - rb[:, 0:2]
        a = to_goal / to_goal.norm(dim=1, keepdim=True).clamp(min=1e-6)
        od, rd, td, ud, idv = dev.step_device(a)
        ended |= (td | ud).cpu().numpy()
        if ended.all():
            break
    assert ended.all()
    seeds1 = dev._dl["seeds"].cpu().numpy()
    assert np.all((seeds1 - seeds0) % W == 0) and np.all(seeds1 > seeds0)
    fresh = (dev._dl["counter"] == 0).cpu().numpy()          # worlds regenerated in the very last step
    assert fresh.any()
    check = BatchedSocialNavGym(cfg, W)
    check.reset(phase="test", first_case=3, device=True)
    from social_navigation_pyenvs_amd.generators import generate_worlds
    generate_worlds(check.cw, "circle_crossing", seeds1.astype(np.uint32), insert_robot=True)
    np.testing.assert_array_equal(dev.cw.get_states()[fresh], check.cw.get_states()[fresh])
    np.testing.assert_array_equal(dev.cw.get_robot()[fresh], check.cw.get_robot()[fresh])
    assert np.all(dev._dl["gtime"].cpu().numpy()[fresh] == 0)


def test_device_step_next_step_autoreset_mode():
    """step_device(auto_reset="next_step") (Gymnasium's NEXT_STEP mode; the next episode comes from the staging batch one step later): until a
    world's first termination it is bit-identical to the same-step mode; a world that ends at step t returns at t + 1 reward 0, not
    terminated, Nothing, a restarted clock, and the first observation of the world its next seed generates."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.generators import generate_worlds
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    W = 64
    cfg = _config("circle_crossing", human_num=6)
    same = BatchedSocialNavGym(cfg, W); same.reset(phase="test", first_case=3, device=True)
    nxt = BatchedSocialNavGym(cfg, W); nxt.reset(phase="test", first_case=3, device=True)
    alive = np.ones(W, bool)                      # worlds that have not terminated yet (in either environment: same inputs)
    prev_done = np.zeros(W, bool)
    seeds_prev = nxt._device_loop_state()["seeds"].cpu().numpy().copy()
    checked = 0
    for k in range(60):
        rb = nxt.cw.d_robot.torch()
        to_goal = rb[:, 10:12] - rb[:, 0:2]
        a = (to_goal / to_goal.norm(dim=1, keepdim=True).clamp(min=1e-6)).contiguous().clone()
        a[::3] *= 0.3                              # a third of the robots dawdle: terminations spread over many steps
        o1, r1, t1, u1, i1 = same.step_device(a)
        o2, r2, t2, u2, i2 = nxt.step_device(a, auto_reset="next_step")
        o1, o2 = o1.cpu().numpy(), o2.cpu().numpy()
        done2 = (t2 | u2).cpu().numpy()
        done1 = (t1 | u1).cpu().numpy()
        # (a) before its first termination a world is the same in both modes (the terminal step included, except the observation,
        #     which same-step mode has already replaced by the new episode's)
        np.testing.assert_array_equal(r1.cpu().numpy()[alive], r2.cpu().numpy()[alive])
        np.testing.assert_array_equal(done1[alive], done2[alive])
        still = alive & ~done2
        np.testing.assert_array_equal(o1[still], o2[still])
        # (b) the step after a termination is the reset step
        if prev_done.any():
            assert np.all(r2.cpu().numpy()[prev_done] == 0) and not done2[prev_done].any() and np.all(i2.cpu().numpy()[prev_done] == 0)
            assert np.all(nxt._dl["gtime"].cpu().numpy()[prev_done] == 0)
            seeds_now = nxt._dl["seeds"].cpu().numpy()
            assert np.all(seeds_now[prev_done] == seeds_prev[prev_done] + W)     # (advanced when the episode ended)
            check = BatchedSocialNavGym(cfg, W); check.reset(phase="test", first_case=3, device=True)
            generate_worlds(check.cw, "circle_crossing", seeds_now.astype(np.uint32), insert_robot=True)
            np.testing.assert_array_equal(nxt.cw.get_states()[prev_done], check.cw.get_states()[prev_done])
            np.testing.assert_array_equal(o2[prev_done], check.observe()[prev_done])
            checked += int(prev_done.sum())
        seeds_prev = np.where(prev_done, nxt._dl["seeds"].cpu().numpy(), seeds_prev)
        alive &= ~done2
        prev_done = done2
    assert checked >= W // 2, checked


def test_device_resident_lookahead_feeds_one_batched_value_network_call():
    """BatchedSocialNavGym.lookahead_device (cs_peek + cs_lookahead on resident tensors) == the host path of
    crowd_nav.policy.cadrl.compute_rotated_states_and_reward on downloaded arrays; then a greedy decision loop entirely on the
    GPU: look-ahead -> one batched (toy) value network call -> arg-max action -> step_device."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.crowd_nav.policy.cadrl import build_action_space_array, compute_rotated_states_and_reward
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    W = 48
    env = BatchedSocialNavGym(_config("circle_crossing", human_num=6), W)
    env.reset(phase="val", first_case=5, device=True)
    acts = build_action_space_array(1.0)
    for _ in range(3):
        env.step_device(torch.as_tensor(np.tile([[0.2, 0.6]], (W, 1)), dtype=torch.float32, device="cuda"))
    rot, rew = env.lookahead_device(acts)
    assert rot.is_cuda and rot.shape == (W, 81, 6, 13) and rew.shape == (W, 81)
    cw = env.cw
    nxt = cw.peek(env.robot_time_step)[:, :, [0, 1, 3, 4]]
    cur = cw.get_states()[:, :6][:, :, [0, 1, 3, 4, 8]]
    rob = cw.get_robot()[:, [0, 1, 3, 4, 8, 10, 11, 12, 2]]
    hrot, hrew = compute_rotated_states_and_reward(acts, nxt, cur, rob, env.robot_time_step)
    np.testing.assert_array_equal(rot.cpu().numpy(), hrot.astype(np.float32))
    np.testing.assert_array_equal(rew.cpu().numpy(), hrew.astype(np.float32))
    # greedy loop on the device: value = reward + a toy network (mean over humans of a linear map of the 13-column rows)
    wts = torch.linspace(-0.05, 0.05, 13, device="cuda")
    acts_d = torch.as_tensor(acts, dtype=torch.float32, device="cuda")
    start = cw.d_robot.torch().view(W, 13)[:, 0:2].clone()
    for _ in range(8):
        rot, rew = env.lookahead_device(acts_d)
        value = rew + 0.9 * (rot @ wts).mean(dim=2)          # ONE batched "model" call over [W, 81, N, 13]
        best = value.argmax(dim=1)
        env.step_device(acts_d.index_select(0, best))
    moved = (cw.d_robot.torch().view(W, 13)[:, 0:2] - start).norm(dim=1)
    assert torch.isfinite(moved).all() and (moved > 0).any()


def test_device_resident_lookahead_with_theta_and_omega_visible():
    """lookahead_device for a policy that sees headings (with_theta_and_omega_visible: 15-column value-network rows, cadrl.py:42-83
    with theta_and_omega_visible=True) == the host function on downloaded arrays, bit for bit."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.crowd_nav.policy.cadrl import build_action_space_array, compute_rotated_states_and_reward
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    W, n = 33, 7
    env = BatchedSocialNavGym(_config("hybrid_scenario", human_num=n), W, headed_obs=True)
    env.reset(phase="val", first_case=11, device=True)
    acts = build_action_space_array(1.0)
    for _ in range(4):
        env.step_device(torch.as_tensor(np.tile([[0.3, 0.5]], (W, 1)), dtype=torch.float32, device="cuda"))
    rot, rew = env.lookahead_device(acts)
    assert rot.is_cuda and rot.shape == (W, 81, n, 15) and rew.shape == (W, 81)
    cw = env.cw
    nxt = cw.peek(env.robot_time_step)[:, :, 0:6]
    cur = cw.get_states()[:, :n][:, :, [0, 1, 3, 4, 8, 2, 7]]
    rob = cw.get_robot()[:, [0, 1, 3, 4, 8, 10, 11, 12, 2]]
    hrot, hrew = compute_rotated_states_and_reward(acts, nxt, cur, rob, env.robot_time_step, True)
    np.testing.assert_array_equal(rot.cpu().numpy(), hrot.astype(np.float32))
    np.testing.assert_array_equal(rew.cpu().numpy(), hrew.astype(np.float32))
    assert np.any(rot.cpu().numpy()[..., 14] != 0)          # omegas of the headed humans really are in the rows


def test_more_than_64_humans_use_the_lane_per_world_kernel():
    """n <= 64 runs one wavefront per world, larger worlds one lane per world: both restate the same stream."""
    from social_navigation_pyenvs_amd.generators import generate_worlds
    from social_navigation_pyenvs_amd.social_gym.social_nav_sim import SocialNavSim

    host = types.SimpleNamespace(_attributes=SocialNavSim._attributes)
    for n, R in ((70, 30), (64, 28)):
        cw = _blank(4, n, 2)
        seeds = np.array([11, 12, 13, 14])
        generate_worlds(cw, "circle_crossing", seeds, circle_radius=R, randomize_attributes=True)
        S = cw.get_states()
        for w in range(4):
            np.random.seed(int(seeds[w]))
            data = SocialNavSim.generate_circular_crossing_setting(
                host, insert_robot=True, human_policy="hsfm_farina", robot_radius=0.3, circle_radius=R, n_actors=n,
                randomize_human_positions=True, randomize_human_attributes=True)
            hp = np.array([data["humans"][i]["pos"] for i in range(n)])
            hr = np.array([data["humans"][i]["radius"] for i in range(n)])
            np.testing.assert_allclose(S[w, :, 0:2], hp.astype(np.float32), **F32_TOL)
            np.testing.assert_array_equal(S[w, :, 8], hr.astype(np.float32))


def test_failed_regeneration_never_replaces_a_live_world():
    """The masked auto-reset of step_device: cs_generate_worlds into a staging batch with a status array, then
    cs_copy_worlds_masked_status -- a world whose bounded rejection sampling gives up (status != 0; forced here with max_tries = 2) keeps its live rows, the others are replaced by the regenerated ones."""
    import ctypes as C

    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd import generators as gen

    W, n = 64, 10
    live, staging = _blank(W, n, 2), _blank(W, n, 2)
    gen.generate_worlds(live, "circle_crossing", 100 + np.arange(W))                   # the live episode
    before = live.get_states().copy()
    g = gen.make_generator(staging, "circle_crossing", max_tries=2)                   # ten humans, two attempts each: some worlds fail
    seeds = _lib.DeviceBuffer.from_numpy((5000 + np.arange(W)).astype(np.uint32), dtype=np.uint32)
    mask = np.zeros(W, np.int32); mask[::2] = 1                                        # every second world's episode ended
    d_mask = _lib.DeviceBuffer.from_numpy(mask, dtype=np.int32)
    d_status = _lib.DeviceBuffer.from_numpy(np.zeros(W, np.int32), dtype=np.int32)
    gen.generate_worlds_device(staging, g, seeds, d_mask, d_status=d_status)
    ds, dd = staging.descriptor(), live.descriptor()
    _lib.check(_lib.load().cs_copy_worlds_masked_status(C.byref(ds), C.byref(dd), C.c_void_p(d_mask.ptr), C.c_void_p(d_status.ptr), C.c_void_p(None)))
    status, after, fresh = d_status.download(), live.get_states(), staging.get_states()
    failed = (status != 0) & (mask != 0)
    replaced = (status == 0) & (mask != 0)
    assert failed.sum() >= 2 and replaced.sum() >= 2, (int(failed.sum()), int(replaced.sum()))
    np.testing.assert_array_equal(after[failed], before[failed])                      # untouched
    np.testing.assert_array_equal(after[mask == 0], before[mask == 0])
    np.testing.assert_array_equal(after[replaced], fresh[replaced])
    assert np.any(after[replaced] != before[replaced])


@pytest.mark.parametrize("next_step", [False, True])
def test_fused_reward_and_bookkeeping_equals_the_two_launches(next_step):
    """cs_collision_reward_gym (the lane that writes a world's reward row also does its episode bookkeeping) leaves exactly what
    cs_collision_reward followed by cs_gym_bookkeeping / cs_gym_bookkeeping_next_step leaves: reward rows, typed results, step counters,
    clocks, reset masks, seeds -- on worlds that collide, reach the goal, time out, are uncomfortable or are being reset."""
    import ctypes as C

    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd.generators import generate_worlds

    W, n = 300, 7
    rng = np.random.default_rng(17 + int(next_step))
    cw = _blank(W, n, 2)
    generate_worlds(cw, "circle_crossing", np.arange(W) + 5, circle_radius=4.0, insert_robot=True)
    rb = cw.get_robot()
    S = cw.get_states()
    rb[: W // 4, 0:2] = S[: W // 4, 2, 0:2] + 0.05             # some robots on top of a human: collision
    rb[W // 4: W // 2, 0:2] = rb[W // 4: W // 2, 10:12] - 0.01  # some next to their goal: ReachGoal
    cw.d_robot.upload(rb.astype(np.float32))
    act = rng.uniform(-0.6, 0.6, (W, 2)).astype(np.float32)
    clock = np.cumsum(np.full(240, 0.25, np.float32), dtype=np.float32)
    clock = np.concatenate([[np.float32(0)], clock]).astype(np.float32)
    counter0 = rng.integers(0, 200, W).astype(np.int32)
    counter0[::9] = 199                                          # ... some at the time limit: Timeout
    gtime0 = clock[counter0]
    seeds0 = (np.arange(W) + 1000).astype(np.uint32)
    prev = (rng.uniform(size=W) < 0.2).astype(np.int32)
    cfg = (C.c_float * 5)(50.0, 1.0, -0.25, 0.2, 0.5)
    lib, d = _lib.load(), cw.descriptor()
    res = []
    for fused in (False, True):
        B = lambda a, t: _lib.DeviceBuffer.from_numpy(np.ascontiguousarray(a), dtype=t)
        b = dict(act=B(act, np.float32), gtime=B(gtime0, np.float32), out=_lib.DeviceBuffer((W, 7)), counter=B(counter0, np.int32),
                 seeds=B(seeds0, np.uint32), mask=B(np.full(W, 7), np.int32), prev=B(prev, np.int32), clock=B(clock, np.float32),
                 reward=_lib.DeviceBuffer((W,)), term=_lib.DeviceBuffer((W,), np.uint8), trunc=_lib.DeviceBuffer((W,), np.uint8),
                 info=_lib.DeviceBuffer((W,), np.int32))
        P = lambda k: C.c_void_p(b[k].ptr)
        if fused:
            book = _lib.cs_gym_book(d_counter=b["counter"].ptr, d_seeds=b["seeds"].ptr, d_mask=b["mask"].ptr,
                                    d_prev_mask=b["prev"].ptr if next_step else None, d_clock=b["clock"].ptr, clock_len=len(clock), auto_reset=1,
                                    d_reward=b["reward"].ptr, d_terminated=b["term"].ptr, d_truncated=b["trunc"].ptr, d_info=b["info"].ptr)
            _lib.check(lib.cs_collision_reward_gym(C.byref(d), P("act"), C.c_float(0.25), P("gtime"), cfg, P("out"), C.byref(book), C.c_void_p(cw.stream)))
        else:
            _lib.check(lib.cs_collision_reward(C.byref(d), P("act"), C.c_float(0.25), P("gtime"), cfg, P("out"), C.c_void_p(cw.stream)))
            if next_step:
                _lib.check(lib.cs_gym_bookkeeping_next_step(C.c_int(W), P("out"), P("counter"), P("seeds"), P("mask"), P("prev"), P("gtime"), P("clock"),
                                                            C.c_int(len(clock)), P("reward"), P("term"), P("trunc"), P("info"), C.c_uint32(0), C.c_void_p(cw.stream)))
            else:
                _lib.check(lib.cs_gym_bookkeeping(C.c_int(W), P("out"), P("counter"), P("seeds"), P("mask"), P("gtime"), P("clock"), C.c_int(len(clock)),
                                                  C.c_int(1), P("reward"), P("term"), P("trunc"), P("info"), C.c_uint32(0), C.c_void_p(cw.stream)))
        cw.sync()
        res.append({k: b[k].download() for k in ("out", "counter", "seeds", "mask", "gtime", "reward", "term", "trunc", "info")})
    for k in res[0]:
        np.testing.assert_array_equal(res[0][k], res[1][k], err_msg=k)
    info = res[1]["info"] if not next_step else res[1]["out"][:, 6].astype(np.int32)
    assert set(np.unique(info)) >= {0, 2, 3, 4}, np.unique(info)   # Nothing, ReachGoal, Collision, Timeout all occurred
    assert res[1]["mask"].sum() > 20


@pytest.mark.parametrize("n,model,robot_row,headed", [(25, "hsfm_farina", False, True), (5, "sfm_helbing", False, False), (10, "hsfm_new_guo", True, True),
                                                      (7, "orca", False, False)])
def test_step_observe_and_copy_observe_equal_the_separate_launches(n, model, robot_row, headed):
    """cs_step_observe (the step kernels write the Gym's observation of the stepped humans from their registers; ORCA: the two launches
    inside the library) == cs_step ; cs_gym_observe, and cs_copy_worlds_masked_observe == cs_copy_worlds_masked_status ; cs_gym_observe on
    the copied worlds -- rows and observations, bit for bit."""
    import ctypes as C

    from social_navigation_pyenvs_amd import _lib, scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W = 37
    rng = np.random.default_rng(n)
    if model == "orca":
        pos, yaw, g = sc.circular_crossing(W, n, 4.0, 77)
        S = sc.make_states(pos, yaw, g).astype(np.float32)
        dd = g[:, :, 0] - S[:, :, 0:2]
        S[:, :, 5:7] = dd / np.linalg.norm(dd, axis=-1, keepdims=True)
        mk = lambda: CrowdWorlds(S, g, None, np.full((W, n), 0.01, np.float32), None, type="orca", robot=np.zeros((W, 13), np.float32))
    else:
        S, g, P, rb = sc.hybrid_worlds(W, n, model, seed0=3)
        R = np.zeros((W, 13), np.float32)
        R[:, 0:2] = rng.uniform(-3, 3, (W, 2)); R[:, 8] = 0.3; R[:, 9] = 80; R[:, 12] = 1.0
        St = np.concatenate([S, R[:, None, :]], axis=1) if robot_row else S
        mk = lambda: CrowdWorlds(St, g, P, None, None, type=model, all_params_equal=True, respawn_bounds=rb,
                                 respawn_worlds=(np.arange(W) % 2 == 1).astype(np.int32), robot_row=robot_row, robot=R)
    act = rng.uniform(-0.5, 0.5, (W, 2)).astype(np.float32)
    lib, Ccols = _lib.load(), (7 if headed else 5)
    a, b = mk(), mk()
    d_act = _lib.DeviceBuffer.from_numpy(act)
    obs_a, obs_b = _lib.DeviceBuffer((W, n, Ccols)), _lib.DeviceBuffer((W, n, Ccols))
    for _ in range(3):
        da, db = a.descriptor(), b.descriptor()
        _lib.check(lib.cs_step(C.byref(da), C.c_float(0.0125), C.c_int(20), C.c_void_p(d_act.ptr), C.c_void_p(a.stream)))
        _lib.check(lib.cs_gym_observe(C.byref(da), C.c_int(int(headed)), C.c_void_p(obs_a.ptr), C.c_void_p(a.stream)))
        _lib.check(lib.cs_step_observe(C.byref(db), C.c_float(0.0125), C.c_int(20), C.c_void_p(d_act.ptr), C.c_int(int(headed)), C.c_void_p(obs_b.ptr),
                                       C.c_void_p(b.stream)))
        a.sync(); b.sync()
        np.testing.assert_array_equal(a.get_states(), b.get_states())
        np.testing.assert_array_equal(obs_a.download(), obs_b.download())
    assert np.abs(obs_a.download()[..., 2:4]).max() > 0.05
    # the masked copy with observation rows
    src = mk()
    mask = (rng.uniform(size=W) < 0.4).astype(np.int32)
    status = np.zeros(W, np.int32); status[np.flatnonzero(mask)[:2]] = 1        # two generations "failed": not copied, rows kept
    d_mask, d_status = _lib.DeviceBuffer.from_numpy(mask, np.int32), _lib.DeviceBuffer.from_numpy(status, np.int32)
    ds, da, db = src.descriptor(), a.descriptor(), b.descriptor()
    _lib.check(lib.cs_copy_worlds_masked_status(C.byref(ds), C.byref(da), C.c_void_p(d_mask.ptr), C.c_void_p(d_status.ptr), C.c_void_p(a.stream)))
    _lib.check(lib.cs_gym_observe(C.byref(da), C.c_int(int(headed)), C.c_void_p(obs_a.ptr), C.c_void_p(a.stream)))
    _lib.check(lib.cs_copy_worlds_masked_observe(C.byref(ds), C.byref(db), C.c_void_p(d_mask.ptr), C.c_void_p(d_status.ptr), C.c_int(int(headed)),
                                                 C.c_void_p(obs_b.ptr), C.c_void_p(b.stream)))
    a.sync(); b.sync()
    np.testing.assert_array_equal(a.get_states(), b.get_states())
    np.testing.assert_array_equal(obs_a.download(), obs_b.download())
    copied = (mask == 1) & (status == 0)
    np.testing.assert_array_equal(b.get_states()[copied], src.get_states()[copied])


def test_step_device_from_the_library_stream_equals_the_cross_stream_call():
    """A loop inside `with torch.cuda.stream(env.device_stream())` (step_device then skips its cross-stream waits) returns what the
    ordinary call returns, auto-resets included."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    W = 96
    cfg = _config("circle_crossing", human_num=6)
    envs = [BatchedSocialNavGym(cfg, W) for _ in range(2)]
    for e in envs:
        e.reset(phase="test", first_case=11, device=True)
    gen = torch.Generator(device="cuda").manual_seed(5)
    acts = [torch.rand(W, 2, device="cuda", generator=gen) * 1.6 - 0.8 for _ in range(40)]
    torch.cuda.synchronize()
    outs = [[], []]
    for a in acts:
        outs[0].append([t.clone() for t in envs[0].step_device(a)])
    with torch.cuda.stream(envs[1].device_stream()):
        for a in acts:
            outs[1].append([t.clone() for t in envs[1].step_device(a)])
    torch.cuda.synchronize()
    ended = 0
    for x, y in zip(*outs):
        for tx, ty in zip(x, y):
            assert torch.equal(tx, ty)
        ended += int((x[2] | x[3]).sum())
    assert ended > 0   # some episodes did end and were regenerated on the way


def test_prestaged_episodes_consume_refill_and_the_in_place_fallback():
    """cs_refill_staged_worlds / cs_consume_staged_worlds at the C ABI (include/crowdstep.h cs_stage_book, depth 2): a finished world
    takes over the episode staged for its next seed -- twice in a row without a refill pass in between; the third time its slot is
    stale and the world is generated in place by the same generator code: the same rows either way, a function of the seed.  A world
    whose generation fails keeps its rows, is flagged, and its sequence moves on; a refill pass regenerates exactly the consumed slots."""
    import ctypes as C

    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd import generators as gen

    W, n, stride, K = 48, 10, 1000, 2
    lib = _lib.load()
    live, ref = _blank(W, n, 2), _blank(W, n, 2)
    staging = live.staging_copy(K)
    assert staging.W == K * W
    base = (100 + np.arange(W)).astype(np.uint32)
    gen.generate_worlds(live, "circle_crossing", base)
    g = gen.make_generator(live, "circle_crossing", max_tries=3)                      # ten humans, three attempts each: some seeds fail
    B = lambda a, t: _lib.DeviceBuffer.from_numpy(np.ascontiguousarray(a), dtype=t)
    b = dict(seeds=B(base, np.uint32), base=B(base, np.uint32), epoch=B(np.zeros(W), np.uint32), staged=B(np.tile(base, K), np.uint32),
             status=B(np.zeros(K * W), np.int32), failed=B(np.full(W, 9), np.int32), obs=_lib.DeviceBuffer((W, n, 5)))
    book = _lib.cs_stage_book(d_seeds=b["seeds"].ptr, d_base_seed=b["base"].ptr, d_epoch=b["epoch"].ptr, d_staged_seed=b["staged"].ptr,
                              d_staged_status=b["status"].ptr, d_failed=b["failed"].ptr, seed_stride=stride, depth=K)
    ds, dd = staging.descriptor(), live.descriptor()
    refill = lambda: (_lib.check(lib.cs_refill_staged_worlds(C.byref(g), C.byref(ds), C.byref(book), C.c_void_p(None))), _lib.stream_sync(None))

    def consume(mask):
        d_mask = B(mask, np.int32)
        _lib.check(lib.cs_consume_staged_worlds(C.byref(g), C.byref(ds), C.byref(dd), C.c_void_p(d_mask.ptr), C.byref(book), C.c_int(0),
                                                C.c_void_p(b["obs"].ptr), C.c_void_p(None)))
        _lib.stream_sync(None)

    def expected(seeds):   # what cs_generate_worlds makes of these seeds with the same generator settings
        st = B(np.zeros(W), np.int32)
        d_seeds = B(seeds, np.uint32)
        dr = ref.descriptor()
        scratch = _lib.DeviceBuffer((int(lib.cs_generate_scratch_bytes(C.c_int(W))) // 4,), np.uint32)
        _lib.check(lib.cs_generate_worlds(C.byref(g), C.byref(dr), C.c_void_p(d_seeds.ptr), C.c_void_p(None), C.c_void_p(st.ptr), C.c_void_p(None),
                                          C.c_void_p(scratch.ptr), C.c_void_p(None)))
        _lib.stream_sync(None)
        return ref.get_states().copy(), ref.get_goals().copy(), ref.get_robot().copy(), st.download()

    E = {e: expected(base + e * stride) for e in (1, 2, 3, 4, 5)}
    assert (E[1][3] != 0).sum() >= 2 and (E[1][3] == 0).sum() >= W // 2, E[1][3]
    refill()                                                                           # first pass: slot 1 <- episode 1, slot 0 <- episode 2
    tags = b["staged"].download().reshape(K, W)
    np.testing.assert_array_equal(tags[1], base + stride); np.testing.assert_array_equal(tags[0], base + 2 * stride)
    st = b["status"].download().reshape(K, W)
    np.testing.assert_array_equal(st[1], E[1][3]); np.testing.assert_array_equal(st[0], E[2][3])
    sS = staging.get_states().reshape(K, W, n, 13)
    np.testing.assert_array_equal(sS[1][E[1][3] == 0], E[1][0][E[1][3] == 0]); np.testing.assert_array_equal(sS[0][E[2][3] == 0], E[2][0][E[2][3] == 0])
    mask = np.zeros(W, np.int32); mask[::3] = 1
    m = mask == 1
    staged_before = staging.get_states().copy()
    for e in (1, 2, 3):            # three ends in a row of every third world, NO refill pass in between: staged, staged, generated in place
        before = live.get_states().copy()
        b["seeds"].upload(base + (e * stride) * mask.astype(np.uint32))              # the bookkeeping has moved the seeds on
        consume(mask)
        S, G, R, stat = E[e]
        ok = stat == 0
        took, kept = m & ok, ~m | ~ok
        assert took.sum() >= 4
        after = live.get_states()
        np.testing.assert_array_equal(after[took], S[took]); np.testing.assert_array_equal(live.get_goals()[took], G[took])
        np.testing.assert_array_equal(live.get_robot()[took], R[took])
        np.testing.assert_array_equal(b["obs"].download()[took], S[took][:, :, [0, 1, 3, 4, 8]])
        np.testing.assert_array_equal(after[kept], before[kept])                       # a failed generation never replaces a live world
        failed = b["failed"].download()
        np.testing.assert_array_equal(failed[m], (~ok[m]).astype(np.int32)); np.testing.assert_array_equal(failed[~m], 9)
        np.testing.assert_array_equal(b["epoch"].download(), e * mask.astype(np.uint32))
        np.testing.assert_array_equal(staging.get_states(), staged_before)             # consuming never writes the staging batch
    # the refill pass regenerates exactly the consumed slots: episodes 4 (slot 0) and 5 (slot 1) of the worlds that ended
    refill()
    tags = b["staged"].download().reshape(K, W)
    np.testing.assert_array_equal(tags[0], base + np.where(m, 4, 2).astype(np.uint32) * stride)
    np.testing.assert_array_equal(tags[1], base + np.where(m, 5, 1).astype(np.uint32) * stride)
    sS = staging.get_states().reshape(K, W, n, 13)
    for slot, e_new, e_old in ((0, 4, 2), (1, 5, 1)):
        redo, left = m & (E[e_new][3] == 0), ~m & (E[e_old][3] == 0)
        np.testing.assert_array_equal(sS[slot][redo], E[e_new][0][redo])
        np.testing.assert_array_equal(sS[slot][left], E[e_old][0][left])              # slots nobody consumed are left alone
    before = b["staged"].download()
    refill()                                                                           # nothing to do: every block leaves at once
    np.testing.assert_array_equal(b["staged"].download(), before)


def test_step_device_is_the_same_whatever_the_refill_cadence():
    """The staging batch is an optimisation, not a semantics: a loop whose refill pass never runs again after the first one (every later
    episode end takes the in-place path) returns what the default cadence returns, bit for bit -- episodes of a few steps included
    (the robots are driven into the nearest human: a world ends, and ends again soon after its reset)."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    W = 80
    cfg = _config("circle_crossing", human_num=6)
    envs = [BatchedSocialNavGym(cfg, W) for _ in range(3)]
    envs[1].REFILL_EVERY, envs[1].STAGE_DEPTH = 10 ** 9, 2
    envs[2].REFILL_EVERY, envs[2].STAGE_DEPTH = 1, 1
    # envs[0] takes the ONE-launch step (cs_gym_step_staged, the default); the two odd cadences run a staging batch dry on purpose and need
    # the in-place generation of the separate consume launch: the two-launch path -- which is thereby also the bit-equality reference of the fold
    envs[1].FOLD_RESET = envs[2].FOLD_RESET = False
    for e in envs:
        e.reset(phase="train", first_case=7, device=True)
    ended = np.zeros(W, int)
    for k in range(90):
        rb = envs[0].cw.d_robot.torch().view(W, 13)
        hum = envs[0].observe_device()[:, :, 0:2]
        d = hum - rb[:, None, 0:2]
        near = d[torch.arange(W), d.norm(dim=2).argmin(dim=1)]
        a = (near / near.norm(dim=1, keepdim=True).clamp(min=1e-6)).contiguous()      # straight at the nearest human
        outs = [[t.clone() for t in e.step_device(a)] for e in envs]
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert torch.equal(x, y), k
        ended += (outs[0][2] | outs[0][3]).cpu().numpy().astype(int)
    assert (ended >= 3).sum() >= W // 4, ended                   # many worlds went through several episodes
    for e in envs[1:]:
        np.testing.assert_array_equal(e.cw.get_states(), envs[0].cw.get_states())
        assert e.failed_resets() == 0
    assert envs[1]._dl["depth"] == 2 and int(envs[1]._dl["epoch"].max()) > 2                   # (its staging batch really ran dry)
    assert envs[0]._dl[("pieces", 0, "same_step")]["fold"] is not None and envs[1]._dl[("pieces", 0, "same_step")]["fold"] is None
    assert int(envs[0]._dl["pending"].sum()) == 0                                                # nothing was ever deferred at the default cadence


@pytest.mark.parametrize("mode", [True, "next_step"])
@pytest.mark.parametrize("W,n,steps,max_tries", [(96, 6, 90, None), (4096, 25, 130, None), (96, 6, 90, 1)])
def test_one_launch_gym_step_equals_the_two_launches(mode, W, n, steps, max_tries):
    """cs_gym_step_staged (reward + bookkeeping, substeps, observation AND the take-over of the staged episodes in one launch) against
    cs_gym_step + cs_consume_staged_worlds, bit for bit: observations, rewards, flags, info codes of every step, the state rows, goal
    lists, robot rows, seeds, epochs and clocks at the end -- same-step and NEXT_STEP rules, a small batch of short episodes (the robots
    run into the nearest human: worlds end again and again) and the benchmark's 4096 x 25 hybrid batch.
    `max_tries`: the generator's bounded rejection sampling gives up on part of the staged episodes (status != 0): such a slot is NOT taken
    over on either path -- the world keeps its stepped rows and a fresh observation, the slot's turn is used up, failed[w] = 1 (round 5's
    one-launch step rewound such a world to its pre-step rows)."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    cfg = _config("circle_crossing" if n == 6 else "hybrid_scenario", human_num=n)
    one, two = BatchedSocialNavGym(cfg, W), BatchedSocialNavGym(cfg, W)
    two.FOLD_RESET = False
    for e in (one, two):
        e.reset(phase="train", first_case=11, device=True)
        if max_tries is not None:
            e._gen_kw = dict(e._gen_kw, max_tries=max_tries)      # (before the device loop's generator is built, at the first step_device)
    failed_seen = 0
    ended = np.zeros(W, int)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    for k in range(steps):
        if n == 6:
            rb = one.cw.d_robot.torch().view(W, 13)
            d = one.observe_device()[:, :, 0:2] - rb[:, None, 0:2]
            near = d[torch.arange(W), d.norm(dim=2).argmin(dim=1)]
            a = (near / near.norm(dim=1, keepdim=True).clamp(min=1e-6)).contiguous()
        else:
            a = torch.randn(W, 2, device="cuda", generator=gen) * 0.7
        o1 = [t.clone() for t in one.step_device(a, auto_reset=mode)]
        o2 = [t.clone() for t in two.step_device(a, auto_reset=mode)]
        for x, y in zip(o1, o2):
            assert torch.equal(x, y), k
        ended += (o1[2] | o1[3]).cpu().numpy().astype(int)
        if max_tries is not None:
            f1, f2 = one.reset_failed_mask(), two.reset_failed_mask()
            assert torch.equal(f1, f2), k
            failed_seen += int((f1 == 1).sum())
    assert one._dl[("pieces", 0, "next_step" if mode == "next_step" else "same_step")]["fold"] is not None
    assert (ended >= 1).sum() >= W // 8, ended.sum()
    np.testing.assert_array_equal(one.cw.get_states(), two.cw.get_states())
    np.testing.assert_array_equal(one.cw.get_goals(), two.cw.get_goals())
    np.testing.assert_array_equal(one.cw.get_robot(), two.cw.get_robot())
    for key in ("seeds", "epoch", "counter", "gtime", "failed"):
        assert torch.equal(one._dl[key], two._dl[key]), key
    assert int(one._dl["pending"].sum()) == 0
    if max_tries is None:
        assert one.failed_resets() == 0
    else:
        fine = int(((one._dl["epoch"] > 0) & (one.reset_failed_mask() == 0)).sum())
        assert failed_seen > 0 and fine > 0, (failed_seen, fine)   # some staged episodes could not be generated, others could


def test_one_launch_gym_step_defers_a_take_over_whose_episode_is_not_staged_yet():
    """A staging batch of ONE episode per world that is refilled only every sixteenth step runs dry when a world ends twice in between.  The
    one-launch step does not generate in place: the take-over is deferred (reset_failed_mask() == 2, pending), the world is between two
    episodes -- reward 0, no flags -- and takes over exactly the episode of its next seed once a refill pass has staged it.  No episode
    is skipped or repeated: every world's epoch equals the number of its episode ends."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.generators import generate_worlds
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    W = 96
    cfg = _config("circle_crossing", human_num=6)
    env = BatchedSocialNavGym(cfg, W)
    env.REFILL_EVERY, env.STAGE_DEPTH = 16, 1
    env.reset(phase="train", first_case=3, device=True)
    ends = np.zeros(W, int)
    deferred_seen = 0
    was_pending = np.zeros(W, bool)
    for k in range(120):
        rb = env.cw.d_robot.torch().view(W, 13)
        d = env.observe_device()[:, :, 0:2] - rb[:, None, 0:2]
        near = d[torch.arange(W), d.norm(dim=2).argmin(dim=1)]
        a = (near / near.norm(dim=1, keepdim=True).clamp(min=1e-6)).contiguous()
        obs, rew, term, trunc, info = [t.clone() for t in env.step_device(a)]
        torch.cuda.synchronize()
        pend = env._dl["pending"].cpu().numpy().astype(bool)
        failed = env.reset_failed_mask().cpu().numpy()
        done = (term | trunc).cpu().numpy()
        # a world that entered this step pending was between two episodes: nothing is reported for it
        assert not done[was_pending].any() and np.all(rew.cpu().numpy()[was_pending] == 0) and np.all(info.cpu().numpy()[was_pending] == 0)
        assert np.all(failed[pend] == 2) and np.all(failed[~pend] == 0)
        ends += done
        deferred_seen += int(pend.sum())
        # a world that took over an episode in this step stands on the first rows of the episode its seed generates
        took = (done | was_pending) & ~pend
        if took.any():
            check = BatchedSocialNavGym(cfg, W); check.reset(phase="train", first_case=3, device=True)
            generate_worlds(check.cw, "circle_crossing", env._dl["seeds"].cpu().numpy().astype(np.uint32), insert_robot=True)
            np.testing.assert_array_equal(env.cw.get_states()[took], check.cw.get_states()[took])
            np.testing.assert_array_equal(obs.cpu().numpy()[took], check.observe()[took])
        was_pending = pend
    assert deferred_seen > 0, "the staging batch never ran dry: the test did not exercise a deferral"
    epoch = env._dl["epoch"].cpu().numpy()
    assert np.array_equal(epoch + was_pending.astype(int), ends), (epoch, ends)          # every ended episode was followed by exactly one take-over
    assert np.array_equal(env._dl["seeds"].cpu().numpy().astype(np.int64) - env._dl["base_seed"].cpu().numpy().astype(np.int64), ends * W)


@pytest.mark.parametrize("n,model,robot_row,next_step,unicycle", [(25, "hsfm_farina", False, False, False), (25, "hsfm_farina", True, True, False), (17, "sfm_guo", False, True, False),
                                                                  (10, "sfm_helbing", False, False, False), (50, "hsfm_new_guo", True, False, False), (7, "orca", False, False, False),
                                                                  (25, "hsfm_farina", True, False, True), (25, "hsfm_farina", False, True, True), (10, "sfm_helbing", False, False, True),
                                                                  (7, "orca", False, False, True)])
def test_gym_step_is_the_head_and_the_body_in_one_launch(n, model, robot_row, next_step, unicycle):
    """cs_gym_step (the step kernel's prologue does the reward / termination of the incoming state and the episode bookkeeping, then the
    fused substeps, then the observation) == cs_collision_reward_gym ; cs_step_observe -- reward rows, typed results, counters, clocks,
    masks, seeds, state rows, goal lists, robot rows and observations, bit for bit, over several steps (so that robots collide, reach
    goals and time out on the way); worlds the LDS kernel does not step (10-row worlds on the DPP-row kernel, ORCA) take the two launches.
    `unicycle`: the action rows are ActionRot (v, r) (CS_ROBOT_UNICYCLE, robot_agent.py:119-136): both heads turn them into the same velocity."""
    import ctypes as C

    from social_navigation_pyenvs_amd import _lib, scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W = 301
    rng = np.random.default_rng(n + int(next_step))
    R = np.zeros((W, 13), np.float32)
    R[:, 0:2] = rng.uniform(-3, 3, (W, 2)); R[:, 8] = 0.3; R[:, 9] = 80; R[:, 10:12] = -R[:, 0:2]; R[:, 12] = 1.0
    if model == "orca":
        pos, yaw, g = sc.circular_crossing(W, n, 4.0, 77)
        S = sc.make_states(pos, yaw, g).astype(np.float32)
        dd = g[:, :, 0] - S[:, :, 0:2]
        S[:, :, 5:7] = dd / np.linalg.norm(dd, axis=-1, keepdims=True)
        mk = lambda: CrowdWorlds(S, g, None, np.full((W, n), 0.01, np.float32), None, type="orca", robot=R)
    else:
        S, g, P, rb = sc.hybrid_worlds(W, n, model, seed0=3)
        R[: W // 5, 0:2] = S[: W // 5, 1, 0:2] + 0.1                                   # some robots on top of a human
        R[W // 5: 2 * W // 5, 0:2] = R[W // 5: 2 * W // 5, 10:12] - 0.01                # some next to their goal
        St = np.concatenate([S, R[:, None, :]], axis=1) if robot_row else S
        mk = lambda: CrowdWorlds(St, g, P, None, None, type=model, all_params_equal=True, respawn_bounds=rb,
                                 respawn_worlds=(np.arange(W) % 2 == 1).astype(np.int32), robot_row=robot_row, robot=R)
    act = rng.uniform(-0.5, 0.5, (W, 2)).astype(np.float32)
    if unicycle:
        act = np.stack([rng.uniform(0.1, 0.9, W), rng.uniform(-0.1, 0.1, W)], -1).astype(np.float32)
        R[:, 2] = rng.uniform(-np.pi, np.pi, W)
        mk0 = mk

        def mk():
            c = mk0()
            c.unicycle = True
            return c
    clock = np.concatenate([[np.float32(0)], np.cumsum(np.full(240, 0.25, np.float32), dtype=np.float32)]).astype(np.float32)
    counter0 = rng.integers(0, 190, W).astype(np.int32); counter0[::9] = 197
    cfg = (C.c_float * 5)(50.0, 1.0, -0.25, 0.2, 0.5)
    lib, Ccols = _lib.load(), 5
    res = []
    for fused in (False, True):
        cw = mk()
        if fused and model != "orca" and n != 10:
            assert "MAXT=64" in cw.step_variant(), cw.step_variant()
        B = lambda a, t: _lib.DeviceBuffer.from_numpy(np.ascontiguousarray(a), dtype=t)
        b = dict(act=B(act, np.float32), gtime=B(clock[counter0], np.float32), out=_lib.DeviceBuffer((W, 7)), counter=B(counter0, np.int32),
                 seeds=B(np.arange(W) + 1000, np.uint32), masks=[B(np.zeros(W), np.int32), B(np.zeros(W), np.int32)], clock=B(clock, np.float32),
                 reward=_lib.DeviceBuffer((W,)), term=_lib.DeviceBuffer((W,), np.uint8), trunc=_lib.DeviceBuffer((W,), np.uint8),
                 info=_lib.DeviceBuffer((W,), np.int32), obs=_lib.DeviceBuffer((W, n, Ccols)))
        P_ = lambda k: C.c_void_p(b[k].ptr)
        hist = []
        for k in range(4):
            d = cw.descriptor()
            p = k & 1
            book = _lib.cs_gym_book(d_counter=b["counter"].ptr, d_seeds=b["seeds"].ptr, d_mask=b["masks"][p].ptr,
                                    d_prev_mask=b["masks"][p ^ 1].ptr if next_step else None, d_clock=b["clock"].ptr, clock_len=len(clock), auto_reset=1,
                                    d_reward=b["reward"].ptr, d_terminated=b["term"].ptr, d_truncated=b["trunc"].ptr, d_info=b["info"].ptr, seed_stride=777)
            if fused:
                _lib.check(lib.cs_gym_step(C.byref(d), C.c_float(0.0125), C.c_int(20), P_("act"), C.c_float(0.25), P_("gtime"), cfg, P_("out"), C.byref(book),
                                           C.c_int(0), P_("obs"), C.c_void_p(cw.stream)))
            else:
                _lib.check(lib.cs_collision_reward_gym(C.byref(d), P_("act"), C.c_float(0.25), P_("gtime"), cfg, P_("out"), C.byref(book), C.c_void_p(cw.stream)))
                _lib.check(lib.cs_step_observe(C.byref(d), C.c_float(0.0125), C.c_int(20), P_("act"), C.c_int(0), P_("obs"), C.c_void_p(cw.stream)))
            cw.sync()
            hist.append({**{kk: b[kk].download() for kk in ("out", "counter", "seeds", "gtime", "reward", "term", "trunc", "info", "obs")},
                         "mask": b["masks"][p].download(), "S": cw.get_states(), "G": cw.get_goals(), "R": cw.get_robot()})
        res.append(hist)
    for k, (x, y) in enumerate(zip(*res)):
        for kk in x:
            np.testing.assert_array_equal(x[kk], y[kk], err_msg=f"step {k}: {kk}")
    infos = np.concatenate([h["out"][:, 6] for h in res[1]]).astype(int)
    assert set(np.unique(infos)) >= ({0, 4} if model == "orca" else {0, 2, 3, 4}), np.unique(infos)
    assert sum(int(h["mask"].sum()) for h in res[1]) > 20


def test_orca_worlds_generated_on_the_device_carry_the_preferred_velocity():
    """An ORCA crowd keeps RVO2's preferred velocity in columns 5:7 (update_goals_orca, motion_model_manager.py:125-133: the unit vector to
    the goal, or the goal offset itself within desired_speed of it), set when the reference builds its simulator.  cs_generate_worlds
    writes it for an ORCA batch -- in the device-resident loop a regenerated world therefore starts its first substep with it, like the
    worlds of reset(): checked on a fresh batch and on worlds the auto-reset regenerated."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.generators import generate_worlds
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    def pref_of(S, n):
        d = S[:, :n, 10:12] - S[:, :n, 0:2]
        dn = np.linalg.norm(d, axis=-1, keepdims=True)
        return np.where(dn > S[:, :n, 12:13], d / np.maximum(dn, np.float32(1e-30)), d).astype(np.float32)

    W, n = 48, 6
    env = BatchedSocialNavGym(_config("hybrid_scenario", human_num=n, policy="orca"), W)
    env.reset(phase="test", first_case=2, device=True)
    assert env.cw.orca
    S0 = env.cw.get_states()
    raw = BatchedSocialNavGym(_config("hybrid_scenario", human_num=n, policy="orca"), W)
    raw.reset(phase="test", first_case=2, device=True)
    generate_worlds(raw.cw, "hybrid_scenario", raw._seeds_host, **raw._gen_kw)          # the generator's own rows, no host fix-up behind them
    Sg = raw.cw.get_states()
    np.testing.assert_array_equal(Sg[:, :n, 5:7], pref_of(Sg, n))
    np.testing.assert_array_equal(Sg, S0)
    assert np.abs(Sg[:, :n, 5:7]).max() > 0.5
    # the device-resident loop: drive the robots to their goals, compare every freshly regenerated world
    seen = 0
    for k in range(60):
        rb = env.cw.d_robot.torch().view(W, 13)
        to_goal = rb[:, 10:12] - rb[:, 0:2]
        a = (to_goal / to_goal.norm(dim=1, keepdim=True).clamp(min=1e-6)).contiguous()
        obs, rew, term, trunc, info = env.step_device(a)
        fresh = (term | trunc).cpu().numpy()
        if fresh.any():
            S = env.cw.get_states()
            np.testing.assert_array_equal(S[fresh][:, :n, 5:7], pref_of(S[fresh], n))
            assert np.all(S[fresh][:, :n, 3:5] == 0)
            seen += int(fresh.sum())
    assert seen >= W // 2, seen


@pytest.mark.parametrize("headed_obs", [False, True])
def test_device_resident_step_with_a_visible_robot_equals_the_host_step(headed_obs):
    """step_device for a Gym whose robot is VISIBLE to the crowd (26th state row, LEAN = 3 build with the Gym head in its prologue) and,
    with headed_obs, 7-column observations: equal to the host-array step of the same batch step by step, and the worlds the auto-reset
    takes over from the staging batch equal the host generator's (robot row included)."""
    torch = pytest.importorskip("torch")
    from social_navigation_pyenvs_amd.generators import generate_worlds
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    W, n = 70, 25
    cfg = _config("hybrid_scenario", human_num=n)
    host = BatchedSocialNavGym(cfg, W, robot_visible=True, headed_obs=headed_obs)
    host.reset(phase="test", first_case=3, device=True)
    dev = BatchedSocialNavGym(cfg, W, robot_visible=True, headed_obs=headed_obs)
    dev.reset(phase="test", first_case=3, device=True)
    assert dev.cw.rows == n + 1 and "LEAN=3" in dev.cw.step_variant()
    rng = np.random.default_rng(1)
    same = np.ones(W, bool)   # worlds no auto-reset has touched yet: there the two paths must agree exactly
    for k in range(5):
        a = rng.uniform(-0.4, 0.4, (W, 2)).astype(np.float32)
        oh, rh, th, uh, ih = host.step(a)
        od, rd, td, ud, idv = dev.step_device(torch.as_tensor(a, device="cuda"))
        assert od.shape == (W, n, 7 if headed_obs else 5)
        np.testing.assert_array_equal(rd.cpu().numpy()[same], rh[same])
        np.testing.assert_array_equal(td.cpu().numpy()[same], th[same])
        np.testing.assert_array_equal(idv.cpu().numpy()[same], ih[same])
        same &= ~(th | uh)
        np.testing.assert_array_equal(od.cpu().numpy()[same], oh[same])
    assert same.sum() > W // 2
    np.testing.assert_array_equal(dev.cw.get_states()[same], host.cw.get_states()[same])
    seen = 0
    for k in range(70):
        rb = dev.cw.d_robot.torch().view(W, 13)
        to_goal = rb[:, 10:12] - rb[:, 0:2]
        a = (to_goal / to_goal.norm(dim=1, keepdim=True).clamp(min=1e-6)).contiguous()
        od, rd, td, ud, idv = dev.step_device(a)
        fresh = (td | ud).cpu().numpy()
        if fresh.any():
            seeds = dev._dl["seeds"].cpu().numpy().astype(np.uint32)
            check = BatchedSocialNavGym(cfg, W, robot_visible=True, headed_obs=headed_obs)
            check.reset(phase="test", first_case=3, device=True)
            generate_worlds(check.cw, "hybrid_scenario", seeds, **check._gen_kw)
            np.testing.assert_array_equal(dev.cw.get_states()[fresh], check.cw.get_states()[fresh])
            np.testing.assert_array_equal(dev.cw.get_robot()[fresh], check.cw.get_robot()[fresh])
            np.testing.assert_array_equal(od.cpu().numpy()[fresh], check.observe()[fresh])
            seen += int(fresh.sum())
    assert seen >= W // 3 and dev.failed_resets() == 0, seen
