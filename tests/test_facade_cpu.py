"""CPU suite for the host-side mirror of the reference interface (no GPU needed): scenario generators
against G6, collision / reward against G5, agent rows, reset() observations against G3."""
import configparser
import types

import numpy as np
import pytest

from golden_io import load_cases


def make_config(human_policy="sfm_helbing", scenario="circle_crossing", human_num=5, robot_visible=False,
                time_step=0.0125, robot_time_step=0.25, randomize=False):
    cfg = configparser.RawConfigParser()
    cfg.read_dict({
        "env": {"time_limit": 50, "time_step": time_step, "robot_time_step": robot_time_step, "val_size": 100,
                "test_size": 500, "randomize_attributes": str(randomize).lower()},
        "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2, "discomfort_penalty_factor": 0.5},
        "sim": {"train_val_sim": scenario, "test_sim": scenario, "traffic_length": 14, "traffic_height": 3,
                "circle_radius": 7, "human_num": human_num},
        "humans": {"visible": "true", "policy": human_policy, "radius": 0.3, "v_pref": 1.0, "sensor": "coordinates"},
        "robot": {"visible": str(robot_visible).lower(), "policy": "none", "radius": 0.3, "v_pref": 1.0, "sensor": "coordinates"},
    })
    return cfg


def make_env(human_policy, scenario, human_num, robot_visible, headed_obs=False):
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import SocialNavGym
    from social_navigation_pyenvs_amd.social_gym.src.robot_agent import RobotAgent

    env = SocialNavGym()
    env.configure(make_config(human_policy, scenario, human_num, robot_visible))
    robot = RobotAgent(env)
    robot.visible, robot.desired_speed, robot.radius, robot.sensor = robot_visible, 1.0, 0.3, "coordinates"
    robot.policy = types.SimpleNamespace(multiagent_training=True, with_theta_and_omega_visible=headed_obs,
                                         kinematics="holonomic", name="fake", query_env=True, time_step=None)
    robot.kinematics = "holonomic"
    env.set_robot(robot)
    return env


def test_generators_reproduce_the_legacy_random_stream_g6():
    from social_navigation_pyenvs_amd.social_gym.social_nav_sim import SocialNavSim

    host = types.SimpleNamespace()
    gens = {"circular_crossing": SocialNavSim.generate_circular_crossing_setting,
            "parallel_traffic": SocialNavSim.generate_parallel_traffic_scenario,
            "circular_crossing_with_static_obstacles": SocialNavSim.generate_circular_crossing_with_static_obstacles,
            "circular_crossing_fixed": SocialNavSim.generate_circular_crossing_setting}
    host._attributes = SocialNavSim._attributes
    checked = 0
    for c in load_cases("g6_generators"):
        name = c["generator"]
        if name == "hybrid_choice":
            for seed, ch in zip(c["seeds"], c["choice"]):
                np.random.seed(seed)
                assert str(np.random.choice(["circle_crossing", "parallel_traffic"])) == ch
            continue
        kw = dict(insert_robot=c["insert_robot"], human_policy="sfm_helbing", headless=True, runge_kutta=False,
                  robot_visible=False, robot_radius=0.3, n_actors=c["n"], randomize_human_attributes=c["randomize_attributes"])
        if name == "parallel_traffic":
            kw.update(traffic_length=14, traffic_height=3)
        else:
            kw.update(circle_radius=7, randomize_human_positions=(name != "circular_crossing_fixed"))
        if c["seed"] >= 0:
            np.random.seed(c["seed"])
        data = gens[name](host, **kw)
        hs = data["humans"]
        n = c["n"]
        assert len(hs) == n
        np.testing.assert_array_equal(np.array([hs[i]["pos"] for i in range(n)]), c["pos"])       # bit-exact
        np.testing.assert_array_equal(np.array([hs[i]["yaw"] for i in range(n)]), c["yaw"])
        np.testing.assert_array_equal(np.array([hs[i]["des_speed"] for i in range(n)]), c["des_speed"])
        np.testing.assert_array_equal(np.array([hs[i]["radius"] for i in range(n)]), c["radius"])
        for i in range(n):
            g = np.array(hs[i]["goals"])
            np.testing.assert_array_equal(g, c["goals"][i][:len(g)])
        if c["insert_robot"]:
            np.testing.assert_array_equal(np.array(data["robot"]["pos"], float), c["robot_pos"])
            np.testing.assert_array_equal(np.array(data["robot"]["goals"], float), c["robot_goals"])
        checked += 1
    assert checked > 100


def test_collision_and_reward_g5():
    from social_navigation_pyenvs_amd.crowd_nav.utils.action import ActionXY
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import SocialNavGym
    from social_navigation_pyenvs_amd.social_gym.src.agent import HumanAgent, RobotAgent

    env = SocialNavGym()
    env.configure(make_config())
    env.robot = RobotAgent(env)
    env.robot.kinematics = "holonomic"
    env.humans = [HumanAgent(env, i, "sfm_helbing", [0.0, 0.0], 0.0, [[1.0, 1.0]]) for i in range(5)]
    for k, c in enumerate(load_cases("g5_reward")):
        for i, h in enumerate(env.humans):
            h.position, h.linear_velocity, h.radius = c["hp"][i].copy(), c["hv"][i].copy(), float(c["hr"][i])
        env.robot.position, env.robot.radius = c["rp"].copy(), c["rr"]
        env.robot.goals = [[float(c["rg"][0]), float(c["rg"][1])]]
        col, dmin, reach = env.collision_detection_and_reaching_goal(ActionXY(float(c["action"][0]), float(c["action"][1])), c["T"])
        reward, term, trunc, info = env.compute_reward_and_infos(col, dmin, reach, c["global_time"], c["T"])
        assert (bool(col), bool(reach)) == (c["collision"], c["reaching_goal"]), k
        assert dmin == c["dmin"] or (np.isinf(dmin) and np.isinf(c["dmin"])) or abs(dmin - c["dmin"]) < 1e-15, k
        assert abs(reward - c["reward"]) < 1e-15 and (term, trunc) == (c["terminated"], c["truncated"]), k
        assert type(info).__name__ == c["info"], k


def test_parameter_rows_match_the_reference_for_every_model():
    from social_navigation_pyenvs_amd.social_gym.src.agent import HumanAgent

    seen = set()
    for c in load_cases("g1_episode"):
        if c["type"] in seen:
            continue
        seen.add(c["type"])
        h = HumanAgent(None, 0, c["model"], [0.0, 0.0], 0.0, [[1.0, 1.0]])
        np.testing.assert_array_equal(h.get_parameters(c["model"]), c["params"][0])
    assert len(seen) == 9


def test_reset_observation_and_packed_arrays_g3():
    """reset() needs no GPU: the first observation, the packed state rows and the respawn switch must equal
    what the reference produced for the same phase / test_case."""
    for k, c in enumerate(load_cases("g3_gym")):
        env = make_env(c["model"], c["scenario"], c["human_num"], c["robot_visible"], c["headed_obs"])
        if c["safety_space"] > 0:
            env.set_safety_space(c["safety_space"])
        ob, info = env.reset(phase=c["phase"], test_case=c["test_case"])
        got = np.array([[o.px, o.py, o.vx, o.vy, o.radius] + ([o.theta, o.omega] if c["headed_obs"] else []) for o in ob])
        np.testing.assert_array_equal(got, c["obs"][0])
        mm = env.motion_model_manager
        np.testing.assert_array_equal(mm.states, c["mm_states"][0])
        np.testing.assert_array_equal(mm.goals, c["mm_goals"][0])
        np.testing.assert_array_equal(mm.safety_space, c["mm_safety"])
        assert mm.parallel_traffic_humans_respawn == c["respawn"]
        assert mm.all_equal_humans == c["all_params_equal"]
        assert type(info[0]).__name__ == "Nothing"
        np.testing.assert_array_equal([*env.robot.position, env.robot.yaw, *env.robot.linear_velocity], c["robot_states"][0])


def test_error_behaviour_matches_the_reference():
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import SocialNavGym

    env = SocialNavGym()
    with pytest.raises(ValueError):
        env.configure(make_config(time_step=0.1, robot_time_step=0.25))   # not a multiple
    with pytest.raises(NotImplementedError):
        env.configure(make_config(human_policy="trajnet"))               # unknown human policy
    env.configure(make_config())
    with pytest.raises(AttributeError):
        env.reset()                                                      # robot has to be set


def test_state_records_flatten_like_the_reference():
    from social_navigation_pyenvs_amd.crowd_nav.utils.state import FullState, JointState, ObservableState

    fs = FullState(1, 2, 3, 4, 0.3, 5, 6, 1.0, 0.5)
    ob = ObservableState(7, 8, 9, 10, 0.3)
    assert fs + ob == (1, 2, 3, 4, 0.3, 5, 6, 1.0, 0.5, 7, 8, 9, 10, 0.3)   # self_state + human_state, as the policies flatten it
    assert str(ob) == "7 8 9 10 0.3"
    JointState(fs, [ob])


def test_robot_state_access_and_robot_motion_model_setup():
    """get_robot_state / set_robot_state (motion_model_manager.py:520-550) and the bookkeeping of set_robot_motion_model
    (:552-589) and of the robot half of set_safety_space (:160-170): host logic, no GPU."""
    env = make_env("sfm_helbing", "circle_crossing", 5, True)
    env.reset(phase="test", test_case=2)
    mm = env.motion_model_manager
    env.robot.linear_velocity[:] = [0.3, -0.2]
    st = mm.get_robot_state()
    np.testing.assert_allclose(st, [*env.robot.position, env.robot.yaw, 0.3, -0.2, 0.0, *env.robot.goals[0]])
    assert mm.get_robot_state(include_goal=False, headed=False).shape == (4,) and mm.get_robot_state(include_goal=False, headed=True).shape == (6,)
    st[0:2] += 1.0
    mm.set_robot_state(st)
    np.testing.assert_allclose(env.robot.position, st[0:2])
    with pytest.raises(Exception, match="does not exist"):
        mm.set_robot_motion_model("nonsense", False)
    mm.set_robot_motion_model("sfm_helbing", True)                # RK45 of the robot (cs_robot_model_rk45)
    assert mm.robot_runge_kutta and not mm.robot_orca
    with pytest.raises(ValueError):
        mm.set_robot_motion_model("orca", True)                   # ORCA is Euler only
    env.set_human_motion_model_as_robot_policy("hsfm_new_guo", False)
    assert mm.robot_motion_model_title == "hsfm_new_guo" and env.robot.headed and not env.robot.orca
    assert env.robot.Ci == 120.0 and env.robot.k_lambda == 0.1    # robot.set_parameters(model)
    mm.set_safety_space(0.2)
    assert abs(env.robot.safety_space - 0.21) < 1e-12 and abs(mm.safety_space[5] - 0.21) < 1e-12
    env.reset(phase="test", test_case=3)                           # the robot keeps its model across resets (social_nav_sim.py:174-179)
    assert env.motion_model_manager is not mm and env.motion_model_manager.robot_motion_model_title == "hsfm_new_guo"
    env.set_human_motion_model_as_robot_policy("orca", False)
    env.motion_model_manager.set_safety_space(0.15)
    assert env.robot.orca and abs(env.motion_model_manager._robot_sim_margin - 0.16) < 1e-12
