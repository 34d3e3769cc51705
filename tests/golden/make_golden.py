#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

TEST INFRASTRUCTURE ONLY.  Run from the repo root:  python tests/golden/make_golden.py [group ...]

The reference (``/root/reference``) is imported through ``_refharness`` (inert stand-ins for
numba / pygame / gymnasium / rvo2 / socialforce); only *data* (inputs and the outputs the
reference produced) is written to the ``.npz`` fixtures.  Groups:

  g1_direct   single ``update_humans_parallel`` call on dense synthetic states (all 9 types,
              walls / robot row / per-agent params / NaN-padded goals / goal switches)
  g1_episode  single calls captured mid-episode through ``MotionModelManager.update_humans``
  g2_block    20 consecutive substeps through ``MotionModelManager.update_humans``
              (circular crossing, parallel traffic with respawn, static obstacles, walls)
  g3_gym      ``SocialNavGym.reset``/``step`` loops with scripted actions
  g4_peek     ``get_next_human_observable_states``
  g5_reward   ``collision_detection_and_reaching_goal`` + ``compute_reward_and_infos``
  g6_generators  scenario generator outputs for fixed seeds
  g7_respawn  parallel-traffic respawn (state just before / after)
  g8_lookahead  compute_rotated_states_and_reward (CADRL / SARL 81-action look-ahead)
  g9_laser    LaserSensor.get_laser_measurements (social_gym/src/sensors.py:51-66)
  g13_block_sizes  20-substep blocks at the row counts of the BASELINE.json configurations (10, 25 traffic, 50, 50 + walls +
              immobile humans), all nine models: the shapes the shape-specialised kernel builds run on
  g15_imitation_rk45  SocialNavGym.imitation_learning_step with the robot's motion model integrated by RK45 (runge_kutta=True)
  g16_policy  CADRL.predict / SARL.predict of the reference (crowd_nav/policy/cadrl.py:235-291, multi_human_rl.py:12-88, sarl.py) with
              seeded weights over Gym episodes of the reference env: weights, joint states, the peeked next human states, the 81 action
              values and the chosen action of every decision (the policy seam, SURVEY.md §8 rows b + f1)
  g17_unicycle  the unicycle robot: RobotAgent.step(ActionRot(v, r), dt) (robot_agent.py:119-136) inside the Gym's substep loop
              (social_nav_gym.py:240-245) run directly on the reference's objects, robot pose + crowd rows per substep; the Gym head for
              that action with robot.theta provided
  g10_social_momentum  MotionModelManager("social_momentum").update_humans single steps (motion_model_manager.py:395-404,
              social_gym/src/social_momentum.py)
"""
from __future__ import annotations

import configparser
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refharness  # noqa: E402
from golden_io import save_cases  # noqa: E402

SFMS = ["sfm_helbing", "sfm_guo", "sfm_moussaid", "hsfm_farina", "hsfm_guo", "hsfm_moussaid",
        "hsfm_new", "hsfm_new_guo", "hsfm_new_moussaid"]
DT = 0.0125

ns = _refharness.import_reference()


# ----------------------------------------------------------------------------- helpers
def default_params(model: str) -> np.ndarray:
    game = types.SimpleNamespace(real_size=15, display_to_real_ratio=1000 / 15)
    from social_gym.src.human_agent import HumanAgent

    h = HumanAgent(game, 0, model, [0.0, 0.0], 0.0, [[1.0, 1.0]])
    return h.get_parameters(model)


def my_walls(rng) -> list:
    """Three polygons (3, 4, 5 vertices, CCW) placed around the origin; our own shapes."""
    out = []
    for k, nv in enumerate((3, 4, 5)):
        c = np.array([math.cos(2.1 * k + 0.3), math.sin(2.1 * k + 0.3)]) * (2.0 + 0.7 * k)
        rad = 0.5 + 0.25 * k
        ang0 = rng.uniform(0, 2 * math.pi)
        verts = []
        for v in range(nv):
            a = ang0 + 2 * math.pi * v / nv
            rr = rad * (0.8 + 0.4 * rng.random())
            verts.append([float(c[0] + rr * math.cos(a)), float(c[1] + rr * math.sin(a))])
        out.append(verts)
    return out


def walls_to_array(walls: list) -> np.ndarray:
    smax = max(len(w) for w in walls)
    arr = np.full((len(walls), smax, 2, 2), np.nan)
    for i, w in enumerate(walls):
        for j in range(len(w)):
            a, b = w[j], w[(j + 1) % len(w)]
            arr[i, j, 0], arr[i, j, 1] = min(a, b), max(a, b)
    return arr


def sample_positions(rng, n, half, min_sep):
    pts = []
    tries = 0
    while len(pts) < n:
        p = rng.uniform(-half, half, size=2)
        tries += 1
        if all(np.linalg.norm(p - q) >= min_sep for q in pts) or tries > 20000:
            pts.append(p)
    return np.array(pts)


# ----------------------------------------------------------------------------- G1 direct
def gen_g1_direct():
    cases = []
    seed = 0
    for t, model in enumerate(SFMS):
        base = default_params(model)
        for n in (5, 10, 25, 50):
            combos = [(w, r, h) for w in (0, 1) for r in (0, 1) for h in (0, 1)]
            if n == 50:
                combos = [(0, 0, 0), (1, 1, 1)]
            for walls_on, robot_on, hetero in combos:
                seed += 1
                rng = np.random.default_rng(10_000 + seed)
                rows = n + robot_on
                half = 0.42 * math.sqrt(rows) + 0.4
                S = np.zeros((rows, 13))
                S[:, 0:2] = sample_positions(rng, rows, half, 0.18)
                S[:, 2] = rng.uniform(-math.pi, math.pi, rows)
                S[:, 3:5] = rng.normal(0, 0.6, (rows, 2))
                S[:, 5:7] = rng.normal(0, 0.6, (rows, 2))
                S[:, 7] = rng.normal(0, 0.8, rows)
                S[:, 8] = rng.uniform(0.2, 0.5, rows)
                S[:, 9] = rng.uniform(60, 90, rows)
                S[:, 12] = rng.uniform(0.5, 1.5, rows)
                if seed % 3 == 0:  # consistent rows, as produced by a previous headed step
                    c, s = np.cos(S[:, 2]), np.sin(S[:, 2])
                    S[:, 3] = c * S[:, 5] - s * S[:, 6]
                    S[:, 4] = s * S[:, 5] + c * S[:, 6]
                G = 3 if seed % 2 else 2
                goals = np.full((n, G, 2), np.nan)
                for i in range(n):
                    k = int(rng.integers(1, G + 1))
                    goals[i, :k] = rng.uniform(-half - 3, half + 3, (k, 2))
                    if rng.random() < 0.25:  # goal inside the radius -> goal switch
                        ang = rng.uniform(0, 2 * math.pi)
                        goals[i, 0] = S[i, 0:2] + 0.7 * S[i, 8] * rng.random() * np.array([math.cos(ang), math.sin(ang)])
                S[:n, 10:12] = goals[:, 0]
                if robot_on:
                    S[n, 10:12] = rng.uniform(-5, 5, 2)
                P = np.tile(base, (n, 1))
                if hetero:
                    nz = base != 0
                    P = P * np.where(nz, rng.uniform(0.8, 1.25, P.shape), 1.0)
                safety = np.zeros(rows) if seed % 4 else np.full(rows, 0.01 + 0.15)
                obstacles = walls_to_array(my_walls(rng)) if walls_on else None
                all_eq = not hetero
                S_in, goals_in = S.copy(), goals.copy()
                out = ns.fp.update_humans_parallel(t, S, goals, obstacles, P, DT, safety, all_eq, bool(robot_on))
                case = dict(type=t, n=n, dt=DT, all_params_equal=all_eq, last_is_robot=bool(robot_on),
                            state_in=S_in, goals_in=goals_in, params=P, safety=safety,
                            state_out=out, goals_out=goals.copy(), state_in_after=S.copy())
                if obstacles is not None:
                    case["obstacles"] = obstacles
                cases.append(case)
    print("g1_direct:", len(cases), "cases ->", save_cases("g1_direct", cases))


# ----------------------------------------------------------------------------- sims
def crossing_config(rng, model, n, radius, walls=None, robot=None, robot_visible=False, attrs=False):
    humans = {}
    pts = []
    for i in range(n):
        while True:
            ang = rng.uniform(0, 2 * math.pi)
            p = radius * np.array([math.cos(ang), math.sin(ang)]) + rng.uniform(-0.5, 0.5, 2)
            if all(np.linalg.norm(p - q) > 0.9 for q in pts):
                break
        pts.append(p)
        humans[i] = {"pos": [float(p[0]), float(p[1])], "yaw": float(ns.utils.bound_angle(math.pi + ang)),
                     "goals": [[float(-p[0]), float(-p[1])], [float(p[0]), float(p[1])]],
                     "des_speed": float(rng.uniform(0.6, 1.4)) if attrs else 1.0,
                     "radius": float(rng.uniform(0.25, 0.45)) if attrs else 0.3}
    data = {"headless": True, "motion_model": model, "runge_kutta": False, "robot_visible": robot_visible,
            "grid": False, "humans": humans, "walls": walls or []}
    if robot is not None:
        data["robot"] = robot
    return data


def record_calls(mm_module):
    """Wrap the array seam so every call's inputs/outputs are captured."""
    rec = []
    orig = ns.fp.update_humans_parallel

    def wrapper(t, S, goals, obstacles, P, dt, safety, all_params_equal=False, last_is_robot=False):
        item = dict(type=int(t), n=int(len(S) - int(last_is_robot)), dt=float(dt),
                    all_params_equal=bool(all_params_equal), last_is_robot=bool(last_is_robot),
                    state_in=S.copy(), goals_in=goals.copy(), params=P.copy(), safety=safety.copy())
        if obstacles is not None:
            item["obstacles"] = obstacles.copy()
        out = orig(t, S, goals, obstacles, P, dt, safety, all_params_equal, last_is_robot)
        item["state_out"] = out.copy()
        item["goals_out"] = goals.copy()
        item["state_in_after"] = S.copy()
        rec.append(item)
        return out

    mm_module.update_humans_parallel = wrapper
    return rec, lambda: setattr(mm_module, "update_humans_parallel", orig)


def gen_g1_episode():
    cases = []
    rec, restore = record_calls(ns.mmm)
    seed = 0
    try:
        for t, model in enumerate(SFMS):
            for n, radius, nsteps in ((5, 2.0, 110), (10, 2.6, 130), (25, 3.6, 150)):
                for walls_on, robot_on in ((0, 0), (1, 1)):
                    seed += 1
                    rng = np.random.default_rng(20_000 + seed)
                    robot = {"pos": [0.3, -0.4], "yaw": 0.3, "radius": 0.3, "goals": [[0.0, 4.0]]} if robot_on else None
                    cfg = crossing_config(rng, model, n, radius, walls=my_walls(rng) if walls_on else None,
                                          robot=robot, robot_visible=bool(robot_on), attrs=bool(seed % 2))
                    sim = ns.sim.SocialNavSim(cfg, scenario="custom_config", parallelize_humans=True)
                    mm = sim.motion_model_manager
                    if seed % 3 == 0:
                        mm.set_safety_space(0.15)
                    rec.clear()
                    for k in range(nsteps):
                        if robot_on:  # robot drifts with a constant velocity, as robot.step does
                            sim.robot.position = sim.robot.position + np.array([0.2, 0.5]) * DT
                            sim.robot.linear_velocity = np.array([0.2, 0.5])
                        mm.update_humans(k * DT, DT)
                    for idx in (nsteps // 2, nsteps - 1):
                        c = dict(rec[idx])
                        c["model"] = model
                        cases.append(c)
    finally:
        restore()
    print("g1_episode:", len(cases), "cases ->", save_cases("g1_episode", cases))


def mm_snapshot(mm):
    d = dict(states=mm.states.copy(), goals=mm.goals.copy(), params=mm.params.copy(),
             safety=mm.safety_space.copy())
    if mm.obstacles is not None:
        d["obstacles"] = mm.obstacles.copy()
    return d


def gen_g2_block():
    cases = []
    seed = 0
    # (a) custom crossings (with/without walls)
    for t, model in enumerate(SFMS):
        for n, radius, warm in ((5, 2.2, 60), (25, 3.8, 90)):
            seed += 1
            rng = np.random.default_rng(30_000 + seed)
            walls = my_walls(rng) if seed % 2 else None
            cfg = crossing_config(rng, model, n, radius, walls=walls)
            sim = ns.sim.SocialNavSim(cfg, scenario="custom_config", parallelize_humans=True)
            mm = sim.motion_model_manager
            for k in range(warm):
                mm.update_humans(0, DT)
            before = mm_snapshot(mm)
            for k in range(20):
                mm.update_humans(0, DT)
            after = mm_snapshot(mm)
            cases.append(dict(kind="crossing", model=model, type=t, n=n, dt=DT, n_substeps=20,
                              all_params_equal=bool(mm.all_equal_humans), respawn=False,
                              **{f"in_{k}": v for k, v in before.items()},
                              out_states=after["states"], out_goals=after["goals"]))
    # (b) reference generators: parallel traffic (respawn) and static obstacles
    for t, model in enumerate(SFMS):
        for scen in ("parallel_traffic", "circular_crossing_with_static_obstacles"):
            seed += 1
            np.random.seed(1000 + seed)
            n = 8
            kw = {"insert_robot": False, "human_policy": model, "headless": True, "runge_kutta": False,
                  "n_actors": n}
            if scen == "parallel_traffic":
                kw.update(traffic_length=14, traffic_height=3)
            else:
                kw.update(circle_radius=7)
            sim = ns.sim.SocialNavSim(kw, scenario=scen, parallelize_humans=True)
            mm = sim.motion_model_manager
            warm = 40 if scen == "parallel_traffic" else 200
            for k in range(warm):
                mm.update_humans(0, DT)
            if scen == "parallel_traffic":
                # push two humans close to their goal so that the 20-step window contains respawns
                for i in (1, 4):
                    mm.states[i, 0] = mm.goals[i, 0, 0] + 3.0 + 0.02 * (i + 1)
                    mm.humans[i].set_state(mm.states[i, 0:8])
            before = mm_snapshot(mm)
            for k in range(20):
                mm.update_humans(0, DT)
            after = mm_snapshot(mm)
            c = dict(kind=scen, model=model, type=t, n=n, dt=DT, n_substeps=20,
                     all_params_equal=bool(mm.all_equal_humans),
                     respawn=bool(mm.parallel_traffic_humans_respawn),
                     **{f"in_{k}": v for k, v in before.items()},
                     out_states=after["states"], out_goals=after["goals"])
            if mm.parallel_traffic_humans_respawn:
                c["respawn_bounds"] = [float(x) for x in mm.respawn_bounds]
            cases.append(c)
            sim.parallel_traffic_humans_respawn = False
    print("g2_block:", len(cases), "cases ->", save_cases("g2_block", cases))


# ----------------------------------------------------------------------------- Gym level
class FakePolicy:
    """Only the attributes SocialNavGym reads (social_nav_gym.py:102,131,139)."""

    def __init__(self, kinematics="holonomic", headed_obs=False):
        self.multiagent_training = True
        self.with_theta_and_omega_visible = headed_obs
        self.kinematics = kinematics
        self.name = "fake"
        self.query_env = True
        self.time_step = None


def make_env(human_policy, scenario, human_num, robot_visible, headed_obs=False, kinematics="holonomic",
             time_step=0.0125, robot_time_step=0.25, randomize=False, circle_radius=7.0):
    cfg = configparser.RawConfigParser()
    cfg.read_dict({
        "env": {"time_limit": 50, "time_step": time_step, "robot_time_step": robot_time_step, "val_size": 100,
                "test_size": 500, "randomize_attributes": str(randomize).lower()},
        "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2,
                   "discomfort_penalty_factor": 0.5},
        "sim": {"train_val_sim": scenario, "test_sim": scenario, "traffic_length": 14, "traffic_height": 3,
                "circle_radius": circle_radius, "human_num": human_num},
        "humans": {"visible": "true", "policy": human_policy, "radius": 0.3, "v_pref": 1.0,
                   "sensor": "coordinates"},
        "robot": {"visible": str(robot_visible).lower(), "policy": "none", "radius": 0.3, "v_pref": 1.0,
                  "sensor": "coordinates"},
    })
    env = ns.gym.SocialNavGym()
    env.configure(cfg)
    robot = ns.robot_agent.RobotAgent(env)
    robot.visible = robot_visible
    robot.desired_speed = 1.0
    robot.radius = 0.3
    robot.sensor = "coordinates"
    robot.policy = FakePolicy(kinematics, headed_obs)
    robot.kinematics = kinematics
    env.set_robot(robot)
    return env, cfg


def ob_to_array(ob):
    rows = []
    for o in ob:
        row = [o.px, o.py, o.vx, o.vy, o.radius]
        if hasattr(o, "theta"):
            row += [o.theta, o.omega]
        rows.append(row)
    return np.array(rows, dtype=np.float64)


def gen_g3_gym():
    cases = []
    seed = 0
    scenarios = ["circle_crossing", "parallel_traffic", "circular_crossing_with_static_obstacles", "hybrid_scenario"]
    for model in ("sfm_helbing", "hsfm_farina", "hsfm_new_guo", "sfm_moussaid"):
        for scen in scenarios:
            for robot_visible in (False, True):
                seed += 1
                rng = np.random.default_rng(40_000 + seed)
                phase = ("test", "val", "train")[seed % 3]
                test_case = int(rng.integers(0, 90))
                headed_obs = bool(seed % 2) and model.startswith("hsfm")
                # unicycle actions cannot go through the reference's step(): it reads robot.theta,
                # which is None/absent (social_nav_sim.py:973) -> only holonomic is recordable
                kin = "holonomic"
                # static-obstacle generator needs n >= 6 (n = 5 puts obstacles 0 and 2 at the same
                # angle and its rejection loop never ends, social_nav_sim.py:391-392)
                hn = 6 if scen == "circular_crossing_with_static_obstacles" else 5
                env, _ = make_env(model, scen, hn, robot_visible, headed_obs, kin)
                if seed % 4 == 0:
                    env.set_safety_space(0.15)
                ob, info = env.reset(phase=phase, test_case=test_case)
                obs = [ob_to_array(ob)]
                actions, rewards, terms, truncs, infos, dmins = [], [], [], [], [], []
                mm_states = [env.motion_model_manager.states.copy()]
                mm_goals = [env.motion_model_manager.goals.copy()]
                robot_states = [np.array([*env.robot.position, env.robot.yaw, *env.robot.linear_velocity])]
                nsteps = 10
                for k in range(nsteps):
                    if kin == "holonomic":
                        g = env.robot.get_goal_position() - env.robot.position
                        a = g / max(np.linalg.norm(g), 1e-9) * 0.8 + rng.normal(0, 0.15, 2)
                        action = ns.action.ActionXY(float(a[0]), float(a[1]))
                        actions.append([action.vx, action.vy])
                    else:
                        action = ns.action.ActionRot(float(rng.uniform(0.3, 1.0)), float(rng.uniform(-0.05, 0.05)))
                        actions.append([action.v, action.r])
                    ob, r, term, trunc, info = env.step(action)
                    obs.append(ob_to_array(ob))
                    rewards.append(float(r)); terms.append(bool(term)); truncs.append(bool(trunc))
                    infos.append(type(info[0]).__name__)
                    dmins.append(float(getattr(info[0], "min_dist", np.nan)))
                    mm_states.append(env.motion_model_manager.states.copy())
                    mm_goals.append(env.motion_model_manager.goals.copy())
                    robot_states.append(np.array([*env.robot.position, env.robot.yaw, *env.robot.linear_velocity]))
                mm = env.motion_model_manager
                cases.append(dict(model=model, scenario=scen, robot_visible=robot_visible, phase=phase,
                                  test_case=test_case, headed_obs=headed_obs, kinematics=kin, human_num=hn,
                                  safety_space=float(env.safety_space),
                                  respawn=bool(mm.parallel_traffic_humans_respawn),
                                  obs=np.array(obs), actions=np.array(actions), rewards=np.array(rewards),
                                  terminated=np.array(terms), truncated=np.array(truncs), infos=infos,
                                  dmins=np.array(dmins), mm_states=np.array(mm_states), mm_goals=np.array(mm_goals),
                                  mm_safety=mm.safety_space.copy(), all_params_equal=bool(mm.all_equal_humans),
                                  robot_states=np.array(robot_states), global_time=float(env.global_time)))
                env.parallel_traffic_humans_respawn = False
    print("g3_gym:", len(cases), "cases ->", save_cases("g3_gym", cases))


def gen_g4_peek():
    cases = []
    seed = 0
    for model in ("sfm_helbing", "sfm_guo", "hsfm_farina", "hsfm_new_moussaid"):
        for scen in ("circle_crossing", "parallel_traffic"):
            for robot_visible in (False, True):
                seed += 1
                env, _ = make_env(model, scen, 6, robot_visible)
                env.reset(phase="test", test_case=seed)
                for k in range(4):
                    env.step(ns.action.ActionXY(0.3, 0.6))
                mm = env.motion_model_manager
                before = mm.states.copy()
                goals_before = mm.goals.copy()
                nxt4 = mm.get_next_human_observable_states(0.25)
                mid = mm.states.copy()
                nxt8 = mm.get_next_human_observable_states(0.25, theta_and_omega_visible=True)
                cases.append(dict(model=model, scenario=scen, robot_visible=robot_visible, test_case=seed,
                                  dt=0.25, states_before=before, goals_before=goals_before,
                                  params=mm.params.copy(), safety=mm.safety_space.copy(),
                                  all_params_equal=bool(mm.all_equal_humans), type=int(mm.sfm_type),
                                  next4=nxt4, next8=nxt8, states_after=mm.states.copy(), states_mid=mid,
                                  goals_after=mm.goals.copy()))
                env.parallel_traffic_humans_respawn = False
    print("g4_peek:", len(cases), "cases ->", save_cases("g4_peek", cases))


def gen_g5_reward():
    rng = np.random.default_rng(50_000)
    env, _ = make_env("sfm_helbing", "circle_crossing", 5, False)
    env.reset(phase="test", test_case=0)
    cases = []
    for k in range(200):
        n = 5
        hp = rng.uniform(-2, 2, (n, 2)) * (0.4 if k % 3 == 0 else 1.0)
        hv = rng.normal(0, 0.7, (n, 2))
        hr = rng.uniform(0.2, 0.5, n)
        rp = rng.uniform(-1, 1, 2)
        rg = rp + rng.uniform(-1, 1, 2) * (0.3 if k % 4 == 0 else 3.0)
        rr = float(rng.uniform(0.2, 0.4))
        a = rng.normal(0, 0.8, 2)
        T = 0.25
        gt = float(rng.choice([0.0, 10.0, 48.9, 49.0, 49.2]))
        for i, h in enumerate(env.humans):
            h.position = hp[i].copy(); h.linear_velocity = hv[i].copy(); h.radius = float(hr[i])
        env.robot.position = rp.copy(); env.robot.radius = rr
        env.robot.goals = [[float(rg[0]), float(rg[1])]]
        col, dmin, reach = env.collision_detection_and_reaching_goal(ns.action.ActionXY(float(a[0]), float(a[1])), T)
        reward, term, trunc, info = env.compute_reward_and_infos(col, dmin, reach, gt, T)
        cases.append(dict(hp=hp, hv=hv, hr=hr, rp=rp, rg=rg, rr=rr, action=a, T=T, global_time=gt,
                          time_limit=50, collision=bool(col), dmin=float(dmin), reaching_goal=bool(reach),
                          reward=float(reward), terminated=bool(term), truncated=bool(trunc),
                          info=type(info).__name__))
    print("g5_reward:", len(cases), "cases ->", save_cases("g5_reward", cases))


def cfg_to_arrays(data):
    hs = data["humans"]
    n = len(hs)
    gmax = max(len(hs[i]["goals"]) for i in range(n))
    goals = np.full((n, gmax, 2), np.nan)
    for i in range(n):
        goals[i, :len(hs[i]["goals"])] = np.array(hs[i]["goals"])
    d = dict(pos=np.array([hs[i]["pos"] for i in range(n)]), yaw=np.array([hs[i]["yaw"] for i in range(n)]),
             goals=goals, des_speed=np.array([hs[i]["des_speed"] for i in range(n)]),
             radius=np.array([hs[i]["radius"] for i in range(n)]))
    if "robot" in data:
        d["robot_pos"] = np.array(data["robot"]["pos"], dtype=float)
        d["robot_goals"] = np.array(data["robot"]["goals"], dtype=float)
        d["robot_yaw"] = float(data["robot"]["yaw"])
    return d


def gen_g6_generators():
    cases = []
    host = types.SimpleNamespace()
    gens = {"circular_crossing": ns.sim.SocialNavSim.generate_circular_crossing_setting,
            "parallel_traffic": ns.sim.SocialNavSim.generate_parallel_traffic_scenario,
            "circular_crossing_with_static_obstacles": ns.sim.SocialNavSim.generate_circular_crossing_with_static_obstacles}
    for seed in list(range(0, 4)) + list(range(1000, 1004)) + list(range(2000, 2004)):
        for name, fn in gens.items():
            for insert_robot in (False, True):
                for attrs in (False, True):
                    if name == "circular_crossing_with_static_obstacles" and attrs:
                        continue
                    n = 7 if seed % 2 else 5
                    if name == "circular_crossing_with_static_obstacles":
                        n = 7 if seed % 2 else 6  # n = 5 never terminates in the reference
                    np.random.seed(seed)
                    kw = dict(insert_robot=insert_robot, human_policy="sfm_helbing", headless=True, runge_kutta=False,
                              robot_visible=False, robot_radius=0.3, n_actors=n, randomize_human_attributes=attrs)
                    if name == "parallel_traffic":
                        kw.update(traffic_length=14, traffic_height=3)
                    else:
                        kw.update(circle_radius=7, randomize_human_positions=True)
                    data = fn(host, **kw)
                    cases.append(dict(generator=name, seed=seed, n=n, insert_robot=insert_robot,
                                      randomize_attributes=attrs, **cfg_to_arrays(data)))
    # hybrid choice for a range of seeds (np.random.choice on a 2-list after seeding)
    choice = []
    for seed in range(0, 64):
        np.random.seed(seed)
        choice.append(str(np.random.choice(["circle_crossing", "parallel_traffic"])))
    cases.append(dict(generator="hybrid_choice", seeds=list(range(0, 64)), choice=choice))
    # non-random circular crossing (rand=False)
    for insert_robot in (False, True):
        data = gens["circular_crossing"](host, insert_robot=insert_robot, human_policy="sfm_helbing", headless=True,
                                         n_actors=6, circle_radius=7, randomize_human_positions=False)
        cases.append(dict(generator="circular_crossing_fixed", seed=-1, n=6, insert_robot=insert_robot,
                          randomize_attributes=False, **cfg_to_arrays(data)))
    print("g6_generators:", len(cases), "cases ->", save_cases("g6_generators", cases))


def gen_g7_respawn():
    cases = []
    seed = 0
    for model in ("sfm_helbing", "hsfm_farina", "sfm_guo"):
        for robot_visible in (False, True):
            seed += 1
            env, _ = make_env(model, "parallel_traffic", 6, robot_visible)
            if seed % 2 == 0:
                env.set_safety_space(0.1)
            env.reset(phase="test", test_case=10 + seed)
            mm = env.motion_model_manager
            for k in range(3):
                env.step(ns.action.ActionXY(0.5, 0.0))
            # move three humans next to their goal so several respawn in the same call (sequential rule)
            for i in (0, 2, 5):
                mm.states[i, 0] = mm.goals[i, 0, 0] + 2.9 + 0.01 * (i + 1)
                mm.humans[i].set_state(mm.states[i, 0:8])
            before = mm_snapshot(mm)
            robot_before = np.array([*env.robot.position, env.robot.radius, env.robot.safety_space])
            mm.update_humans(0.0, DT)
            after = mm_snapshot(mm)
            cases.append(dict(model=model, type=int(mm.sfm_type), robot_visible=robot_visible, dt=DT,
                              all_params_equal=bool(mm.all_equal_humans),
                              respawn_bounds=[float(x) for x in mm.respawn_bounds], robot=robot_before,
                              human_safety=np.array([h.safety_space for h in mm.humans], dtype=float),
                              **{f"in_{k}": v for k, v in before.items()},
                              out_states=after["states"], out_goals=after["goals"]))
            env.parallel_traffic_humans_respawn = False
    print("g7_respawn:", len(cases), "cases ->", save_cases("g7_respawn", cases))


def gen_g8_lookahead():
    """compute_rotated_states_and_reward (crowd_nav/policy/cadrl.py:42-83) on random robot / human states with the
    81-action holonomic action space of CADRL.build_action_space (cadrl.py:181-206)."""
    import itertools
    from crowd_nav.policy.cadrl import compute_rotated_states_and_reward

    rng = np.random.default_rng(80_000)
    v_pref = 1.0
    speeds = [(np.exp((i + 1) / 5) - 1) / (np.e - 1) * v_pref for i in range(5)]
    rotations = np.linspace(0, 2 * np.pi, 16, endpoint=False)
    actions = np.array([[0.0, 0.0]] + [[s * np.cos(r), s * np.sin(r)] for r, s in itertools.product(rotations, speeds)])
    cases = []
    for k in range(24):
        n = int(rng.choice([1, 5, 10, 25]))
        headed = bool(k % 2)
        robot = np.array([*rng.uniform(-3, 3, 2), *rng.normal(0, 0.5, 2), 0.3, *rng.uniform(-6, 6, 2), v_pref, 0.0])
        if k % 5 == 0:  # goal within reach of some action
            robot[5:7] = robot[0:2] + rng.uniform(-0.2, 0.2, 2)
        cur = np.zeros((n, 7 if headed else 5))
        cur[:, 0:2] = robot[0:2] + rng.uniform(-2.5, 2.5, (n, 2)) * (0.4 if k % 3 == 0 else 1.0)
        cur[:, 2:4] = rng.normal(0, 0.6, (n, 2))
        cur[:, 4] = rng.uniform(0.25, 0.45, n)
        if headed:
            cur[:, 5] = rng.uniform(-np.pi, np.pi, n); cur[:, 6] = rng.normal(0, 0.5, n)
        dt = 0.25
        nxt = np.zeros((n, 6 if headed else 4))
        drift = rng.normal(0, 0.05, (n, 2))
        if headed:
            nxt[:, 0:2] = cur[:, 0:2] + cur[:, 2:4] * dt + drift; nxt[:, 2] = cur[:, 5] + cur[:, 6] * dt
            nxt[:, 3:5] = cur[:, 2:4] + drift; nxt[:, 5] = cur[:, 6]
        else:
            nxt[:, 0:2] = cur[:, 0:2] + cur[:, 2:4] * dt + drift; nxt[:, 2:4] = cur[:, 2:4] + drift
        rot, rew = compute_rotated_states_and_reward(actions, nxt, cur, robot, dt, theta_and_omega_visible=headed)
        cases.append(dict(n=n, headed=headed, dt=dt, actions=actions, robot=robot, current=cur, next=nxt,
                          rotated=rot, rewards=rew))
    print("g8_lookahead:", len(cases), "cases ->", save_cases("g8_lookahead", cases))


def gen_g9_laser():
    """LaserSensor.get_laser_measurements on random robot poses, human discs and polygon walls (sensors.py:24-66)."""
    from social_gym.src.sensors import LaserSensor

    rng = np.random.default_rng(90_000)
    game = types.SimpleNamespace(real_size=15, display_to_real_ratio=1000 / 15)
    cases = []
    for k in range(40):
        n = int(rng.choice([0, 1, 5, 10, 25]))
        with_walls = k % 2 == 1
        samples = int(rng.choice([1, 2, 31, 61, 181])) if k > 3 else [1, 2, 61, 360][k]
        rng_angle = float(rng.choice([math.pi / 2, math.pi, 2 * math.pi]))
        max_distance = float(rng.choice([4.0, 8.0, 10.0]))
        pos = rng.uniform(-2.5, 2.5, 2)
        yaw = float(rng.uniform(-math.pi, math.pi))
        hp = pos + rng.uniform(-6, 6, (n, 2))
        hr = rng.uniform(0.2, 1.0, n)
        if n and k % 5 == 0:
            hp[0] = pos + 0.1  # the sensor stands inside a disc: t < 0 -> max_distance (sensors.py:32)
        humans = [types.SimpleNamespace(position=hp[i].copy(), radius=float(hr[i])) for i in range(n)]
        verts = my_walls(rng) if with_walls else []
        from social_gym.src.obstacle import Obstacle
        walls = [Obstacle(game, v) for v in verts] if with_walls else []
        # uncertainty=None leaves the attribute unset in the reference (sensors.py:15) and get_laser_measurements then
        # raises AttributeError (:65); the noise-free path is reached by setting the attribute from outside
        sigma = 0.05 if k % 4 == 3 else None
        laser = LaserSensor(pos.copy(), yaw, rng_angle, samples, max_distance, uncertainty=sigma)
        if sigma is None:
            laser.uncertainty = None
        noise_seed = 900 + k
        np.random.seed(noise_seed)
        out = laser.get_laser_measurements(humans, walls)
        cases.append(dict(n=n, samples=samples, range=rng_angle, max_distance=max_distance, pos=pos, yaw=yaw,
                          uncertainty=sigma, noise_seed=noise_seed,
                          human_pos=hp.reshape(n, 2), human_radius=hr,
                          obstacles=walls_to_array(verts) if with_walls else np.zeros((0, 1, 2, 2)),
                          angles=np.array(list(out.keys()), dtype=float), measurements=np.array(list(out.values()), dtype=float)))
    print("g9_laser:", len(cases), "cases ->", save_cases("g9_laser", cases))


def gen_g10_social_momentum():
    """Single update_humans(t, dt) calls of the social-momentum crowd model on dense crossings: with / without a visible
    robot, uniform / per-human radius and speed, a safety space, two substep sizes."""
    cases = []
    seed = 0
    for n, radius in ((2, 1.2), (5, 2.0), (10, 2.6), (25, 3.8)):
        for robot_visible in (False, True):
            for attrs in (False, True):
                seed += 1
                rng = np.random.default_rng(100_000 + seed)
                robot = None
                if robot_visible:
                    robot = {"pos": [float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))], "yaw": 0.0, "radius": 0.3,
                             "goals": [[4.0, 4.0]]}
                cfg = crossing_config(rng, "social_momentum", n, radius, robot=robot, robot_visible=robot_visible, attrs=attrs)
                sim = ns.sim.SocialNavSim(cfg, scenario="custom_config", parallelize_humans=False)
                mm = sim.motion_model_manager
                if seed % 3 == 0:  # set_safety_space raises for this model (:159); the filter reads the attribute (:22)
                    for h in mm.humans:
                        h.safety_space = 0.05
                if robot_visible:
                    sim.robot.linear_velocity = np.array([0.4, -0.3])
                dt = DT if seed % 2 else 0.25
                for k in range(int(round((0.6 * radius) / dt))):      # walk into the dense middle
                    mm.update_humans(0, dt)
                for rep in range(6):
                    for k in range(int(round(0.5 / dt)) if rep else 0):
                        mm.update_humans(0, dt)
                    gmax = max(len(h.goals) for h in mm.humans)
                    def snap():
                        g = np.full((n, gmax, 2), np.nan)
                        for i, h in enumerate(mm.humans):
                            g[i, :len(h.goals)] = np.array(h.goals, dtype=float)
                        return dict(pos=np.array([h.position for h in mm.humans], dtype=float),
                                    vel=np.array([h.linear_velocity for h in mm.humans], dtype=float), goals=g)
                    before = snap()
                    rb = None
                    if robot_visible:
                        rb = np.array([*sim.robot.position, *sim.robot.linear_velocity, sim.robot.radius, sim.robot.safety_space], dtype=float)
                    mm.update_humans(0, dt)
                    after = snap()
                    cases.append(dict(n=n, dt=dt, robot_visible=robot_visible,
                                      radius=np.array([h.radius for h in mm.humans], dtype=float),
                                      safety=np.array([h.safety_space for h in mm.humans], dtype=float),
                                      vd=np.array([h.desired_speed for h in mm.humans], dtype=float),
                                      robot=rb if rb is not None else np.zeros(0),
                                      **{f"in_{k}": v for k, v in before.items()}, **{f"out_{k}": v for k, v in after.items()}))
    print("g10_social_momentum:", len(cases), "cases ->", save_cases("g10_social_momentum", cases))


def robot_row(robot):
    return np.array([*robot.position, robot.yaw, *robot.linear_velocity, *robot.body_velocity, robot.angular_velocity,
                     robot.radius, robot.mass, *robot.goals[0], robot.desired_speed, robot.safety_space,
                     *np.asarray(robot.desired_force, dtype=float)], dtype=np.float64)


def gen_g11_imitation():
    """The robot driven by a human motion model (set_human_motion_model_as_robot_policy + imitation_learning_step,
    social_nav_gym.py:252-274, motion_model_manager.py:552-653).  Two families:
      gym   - SocialNavGym episodes of imitation_learning_step() for the nine SFM / HSFM robot models (no walls in the Gym
              scenarios), robot visible or not, with / without a safety space;
      walls - manager level: update_robot + update_humans substeps in a custom scene with walls (obstacle force of the robot).
    The ORCA robot model needs rvo2 (absent) and is not recordable."""
    cases = []
    seed = 0
    human_models = ["sfm_helbing", "hsfm_new_guo", "hsfm_farina", "sfm_moussaid", "sfm_guo"]
    scenarios = ["circle_crossing", "hybrid_scenario", "parallel_traffic"]
    for rmodel in SFMS:
        for robot_visible in (False, True):
            seed += 1
            rng = np.random.default_rng(110_000 + seed)
            hmodel = human_models[seed % len(human_models)]
            scen = scenarios[seed % len(scenarios)]
            env, _ = make_env(hmodel, scen, 5, robot_visible, headed_obs=bool(seed % 2) and hmodel.startswith("hsfm"))
            env.set_human_motion_model_as_robot_policy(rmodel, False)
            if seed % 3 == 0:
                env.set_safety_space(0.1)
            phase = ("test", "val", "train")[seed % 3]
            test_case = int(rng.integers(0, 90))
            ob, info = env.reset(phase=phase, test_case=test_case)
            mm = env.motion_model_manager
            obs, rewards, terms, truncs, infos, dmins = [ob_to_array(ob)], [], [], [], [], []
            robots = [robot_row(env.robot)]
            mm_states, mm_goals = [mm.states.copy()], [mm.goals.copy()]
            for k in range(24):
                ob, r, term, trunc, info = env.imitation_learning_step()
                obs.append(ob_to_array(ob)); rewards.append(float(r)); terms.append(bool(term)); truncs.append(bool(trunc))
                infos.append(type(info[0]).__name__)
                dmins.append(float(getattr(info[0], "min_dist", np.nan)))
                robots.append(robot_row(env.robot))
                mm_states.append(mm.states.copy()); mm_goals.append(mm.goals.copy())
            cases.append(dict(family="gym", robot_model=rmodel, model=hmodel, scenario=scen, robot_visible=robot_visible,
                              phase=phase, test_case=test_case, human_num=5, safety_space=float(env.safety_space),
                              headed_obs=bool(seed % 2) and hmodel.startswith("hsfm"),
                              respawn=bool(mm.parallel_traffic_humans_respawn),
                              obs=np.array(obs), rewards=np.array(rewards), terminated=np.array(terms),
                              truncated=np.array(truncs), infos=infos, dmins=np.array(dmins), robots=np.array(robots),
                              mm_states=np.array(mm_states), mm_goals=np.array(mm_goals), mm_safety=mm.safety_space.copy(),
                              human_safety=np.array([h.safety_space for h in mm.humans], dtype=float),
                              robot_params=env.robot.get_parameters(rmodel), global_time=float(env.global_time)))
            env.parallel_traffic_humans_respawn = False
    for rmodel in SFMS:
        for robot_visible in (False, True):
            seed += 1
            rng = np.random.default_rng(110_000 + seed)
            hmodel = human_models[seed % len(human_models)]
            n = int(rng.integers(3, 9))
            walls = my_walls(rng)
            robot = {"pos": [float(rng.uniform(-4.5, -3.5)), float(rng.uniform(-1, 1))], "yaw": float(rng.uniform(-0.5, 0.5)),
                     "radius": 0.3, "goals": [[float(rng.uniform(3.5, 4.5)), float(rng.uniform(-1, 1))]]}
            cfg = crossing_config(rng, hmodel, n, 3.5, walls=walls, robot=robot, robot_visible=robot_visible, attrs=bool(seed % 2))
            sim = ns.sim.SocialNavSim(cfg, scenario="custom_config", parallelize_humans=True)
            sim.set_human_motion_model_as_robot_policy(rmodel, False)
            mm = sim.motion_model_manager
            if seed % 3 == 0:
                mm.set_safety_space(0.07)
            robots = [robot_row(sim.robot)]
            mm_states, mm_goals = [mm.states.copy()], [mm.goals.copy()]
            nsub = 200
            for k in range(nsub):
                mm.update_robot(k * DT, DT)
                mm.update_humans(k * DT, DT)
                robots.append(robot_row(sim.robot))
                if (k + 1) % 20 == 0:
                    mm_states.append(mm.states.copy()); mm_goals.append(mm.goals.copy())
            cases.append(dict(family="walls", robot_model=rmodel, model=hmodel, robot_visible=robot_visible, n=n,
                              walls=walls_to_array(walls), robots=np.array(robots), mm_states=np.array(mm_states),
                              mm_goals=np.array(mm_goals), mm_safety=mm.safety_space.copy(), mm_params=mm.params.copy(),
                              human_safety=np.array([h.safety_space for h in mm.humans], dtype=float),
                              all_params_equal=bool(mm.all_equal_humans),
                              robot_params=sim.robot.get_parameters(rmodel), nsub=nsub))
    print("g11_imitation:", len(cases), "cases ->", save_cases("g11_imitation", cases))


def humans_snapshot(mm):
    hs = mm.humans
    gmax = max(len(h.goals) for h in hs)
    g = np.full((len(hs), gmax, 2), np.nan)
    for i, h in enumerate(hs):
        g[i, :len(h.goals)] = np.array(h.goals, dtype=float)
    rows = np.array([[*h.position, h.yaw, *h.linear_velocity, *h.body_velocity, h.angular_velocity, h.radius, h.mass,
                      *h.goals[0], h.desired_speed, h.safety_space, *np.asarray(h.desired_force, dtype=float)] for h in hs], dtype=float)
    return rows, g


def gen_g12_rk45():
    """MotionModelManager(runge_kutta=True).update_humans(t, dt) (motion_model_manager.py:374-384, 500-550): scipy's RK45 around
    the single-agent force functions, whose right-hand side clamps velocities, switches goals and keeps stale desired forces.
    Records rows before / after every call and the number of right-hand-side evaluations (pins the step-size control)."""
    cases = []
    seed = 0
    for t, model in enumerate(SFMS):
        for n, radius, dt, calls in ((4, 2.0, DT, 40), (9, 2.6, 0.25, 10), (16, 3.2, DT, 30)):
            seed += 1
            rng = np.random.default_rng(120_000 + seed)
            walls = my_walls(rng) if seed % 2 else None
            robot_visible = seed % 3 == 0
            robot = None
            if robot_visible:
                robot = {"pos": [float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))], "yaw": 0.0, "radius": 0.3, "goals": [[4.0, 4.0]]}
            cfg = crossing_config(rng, model, n, radius, walls=walls, robot=robot, robot_visible=robot_visible, attrs=bool(seed % 4 == 1))
            cfg["runge_kutta"] = True
            sim = ns.sim.SocialNavSim(cfg, scenario="custom_config", parallelize_humans=False)
            mm = sim.motion_model_manager
            assert mm.runge_kutta
            if seed % 5 == 0:
                mm.set_safety_space(0.05)
            if robot_visible:
                sim.robot.linear_velocity = np.array([0.3, 0.2])
            counter = {"n": 0}
            for name in ("f_rk45_headed", "f_rk45_not_headed"):
                orig = getattr(mm, name)
                def wrapped(tt, y, _o=orig):
                    counter["n"] += 1
                    return _o(tt, y)
                setattr(mm, name, wrapped)
            warm = int(round(0.45 * radius / dt))
            for k in range(warm):
                mm.update_humans(k * dt, dt)
            rows, goals, nfev = [], [], []
            r0, g0 = humans_snapshot(mm)
            rows.append(r0); goals.append(g0)
            for k in range(calls):
                counter["n"] = 0
                mm.update_humans((warm + k) * dt, dt)
                r1, g1 = humans_snapshot(mm)
                rows.append(r1); goals.append(g1); nfev.append(counter["n"])
            rb = np.zeros(0)
            if robot_visible:
                rb = np.array([*sim.robot.position, *sim.robot.linear_velocity, sim.robot.radius, sim.robot.safety_space], dtype=float)
            cases.append(dict(model=model, type=t, n=n, dt=dt, robot_visible=robot_visible, all_params_equal=bool(mm.all_equal_humans),
                              walls=walls_to_array(walls) if walls else np.zeros((0, 1, 2, 2)), rows=np.array(rows), goals=np.array(goals),
                              nfev=np.array(nfev), robot=rb, params=np.array([h.get_parameters(model) for h in mm.humans])))
    print("g12_rk45:", len(cases), "cases ->", save_cases("g12_rk45", cases))


def gen_g14_rk45_more():
    """The rest of the RK45 surface:
      complete - MotionModelManager.complete_rk45_simulation(0, dt, final_time) (motion_model_manager.py:461-498): one solve with
                 t_eval, the returned human_states, the humans as the solve leaves them, the number of RHS evaluations;
      respawn  - update_humans of an RK45 crowd in a parallel-traffic scene with the respawn rule armed (:374-384 + :405-422);
      robot    - update_robot(t, dt) of a robot whose SFM / HSFM model is integrated with RK45 (:631-640, :661-687), humans standing."""
    cases = []
    seed = 0

    def count_rhs(mm, names):
        counter = {"n": 0}
        for name in names:
            orig = getattr(mm, name)
            def wrapped(tt, y, _o=orig):
                counter["n"] += 1
                return _o(tt, y)
            setattr(mm, name, wrapped)
        return counter

    # ---- complete_rk45_simulation
    for t, model in enumerate(SFMS):
        for n, radius, dt, final in ((4, 2.0, 0.25, 2.0), (8, 2.6, 0.1, 1.5)):
            seed += 1
            rng = np.random.default_rng(140_000 + seed)
            walls = my_walls(rng) if seed % 2 else None
            cfg = crossing_config(rng, model, n, radius, walls=walls, attrs=bool(seed % 4 == 1))
            cfg["runge_kutta"] = True
            sim = ns.sim.SocialNavSim(cfg, scenario="custom_config", parallelize_humans=False)
            mm = sim.motion_model_manager
            counter = count_rhs(mm, ("f_rk45_headed", "f_rk45_not_headed"))
            for k in range(int(round(0.3 * radius / 0.0125))):     # walk in: distinct velocities, some interaction
                mm.update_humans(k * 0.0125, 0.0125)
            r0, g0 = humans_snapshot(mm)
            counter["n"] = 0
            hs = mm.complete_rk45_simulation(0.0, dt, final)
            r1, g1 = humans_snapshot(mm)
            cases.append(dict(family="complete", model=model, type=t, n=n, dt=dt, final_time=final, all_params_equal=bool(mm.all_equal_humans),
                              walls=walls_to_array(walls) if walls else np.zeros((0, 1, 2, 2)), rows0=r0, goals0=g0, rows1=r1, goals1=g1,
                              human_states=np.asarray(hs, dtype=float), nfev=counter["n"],
                              params=np.array([h.get_parameters(model) for h in mm.humans])))
    # ---- RK45 + respawn (parallel traffic through the Gym's scenario generator, manager in RK45 mode)
    for model in ("sfm_helbing", "hsfm_farina", "sfm_guo", "hsfm_new_guo"):
        for robot_visible in (False, True):
            seed += 1
            env, _ = make_env(model, "parallel_traffic", 6, robot_visible)
            env.reset(phase="test", test_case=20 + seed)
            mm = env.motion_model_manager
            mm.runge_kutta = True
            mm.parallel = False
            counter = count_rhs(mm, ("f_rk45_headed", "f_rk45_not_headed"))
            for k in range(8):
                mm.update_humans(k * DT, DT)
            for i in (1, 3, 4):          # three humans on the edge of the 3 m zone: they respawn in the next calls, two in one call
                mm.humans[i].position[0] = mm.humans[i].goals[0][0] + 3.0 + (0.002 if i < 4 else 0.02)
            rows, goals, nfev = [], [], []
            r0, g0 = humans_snapshot(mm)
            rows.append(r0); goals.append(g0)
            for k in range(6):
                counter["n"] = 0
                mm.update_humans((8 + k) * DT, DT)
                r1, g1 = humans_snapshot(mm)
                rows.append(r1); goals.append(g1); nfev.append(counter["n"])
            rb = np.array([*env.robot.position, *env.robot.linear_velocity, env.robot.radius, env.robot.safety_space], dtype=float)
            cases.append(dict(family="respawn", model=model, type=SFMS.index(model), n=6, dt=DT, robot_visible=robot_visible,
                              all_params_equal=bool(mm.all_equal_humans), respawn_bounds=[float(x) for x in mm.respawn_bounds],
                              rows=np.array(rows), goals=np.array(goals), nfev=np.array(nfev), robot=rb,
                              params=np.array([h.get_parameters(model) for h in mm.humans])))
            env.parallel_traffic_humans_respawn = False
    # ---- the robot under RK45
    human_models = ["sfm_helbing", "hsfm_new_guo", "hsfm_farina", "sfm_moussaid", "sfm_guo"]
    for rmodel in SFMS:
        for with_walls in (False, True):
            seed += 1
            rng = np.random.default_rng(140_000 + seed)
            hmodel = human_models[seed % len(human_models)]
            n = int(rng.integers(3, 9))
            walls = my_walls(rng) if with_walls else None
            robot = {"pos": [float(rng.uniform(-1.5, -0.5)), float(rng.uniform(-1, 1))], "yaw": float(rng.uniform(-0.5, 0.5)),
                     "radius": 0.3, "goals": [[float(rng.uniform(3.5, 4.5)), float(rng.uniform(-1, 1))]]}
            cfg = crossing_config(rng, hmodel, n, 2.2, walls=walls, robot=robot, robot_visible=False, attrs=bool(seed % 2))
            sim = ns.sim.SocialNavSim(cfg, scenario="custom_config", parallelize_humans=True)
            sim.set_human_motion_model_as_robot_policy(rmodel, True)
            mm = sim.motion_model_manager
            assert mm.robot_runge_kutta
            if seed % 3 == 0:
                mm.set_safety_space(0.07)
            for k in range(30):              # the crowd walks in (Euler), the robot stands
                mm.update_humans(k * DT, DT)
            counter = count_rhs(mm, ("f_rk45_robot_headed", "f_rk45_robot_not_headed"))
            dt = DT if seed % 2 else 0.25
            robots, nfev = [robot_row(sim.robot)], []
            for k in range(12):
                counter["n"] = 0
                mm.update_robot(k * dt, dt)
                robots.append(robot_row(sim.robot)); nfev.append(counter["n"])
            cases.append(dict(family="robot", robot_model=rmodel, model=hmodel, n=n, dt=dt, walls=walls_to_array(walls) if walls else np.zeros((0, 1, 2, 2)),
                              mm_states=mm.states.copy(), human_safety=np.array([h.safety_space for h in mm.humans], dtype=float),
                              robots=np.array(robots), nfev=np.array(nfev), robot_params=sim.robot.get_parameters(rmodel)))
    print("g14_rk45_more:", len(cases), "cases ->", save_cases("g14_rk45_more", cases))


# ----------------------------------------------------------------------------- G13 blocks at the BASELINE.json sizes
def gen_g13_block_sizes():
    """20-substep blocks (= one Gym step) through MotionModelManager.update_humans at the row counts of the BASELINE.json
    configurations, all nine models, all_equal_humans worlds with goal lists of two entries -- the shapes the shape-specialised
    builds of the step kernel run on (SURVEY.md §8c asks for N in {5, 10, 25, 50}; G2 holds 5, 8 and 25):
      n10        10 humans, dense crossing                                   (configs[1]: 4096 x 10 SFM circle crossing)
      n25_traffic 25 humans, the reference's parallel-traffic generator with respawns inside the window   (configs[2], odd worlds)
      n50        50 humans, dense crossing                                   (configs[4] without obstacles)
      n50_walls_static  50 humans of which 3 immobile (desired speed 0, radius 0.8, goal = own position, as
                 circular_crossing_with_static_obstacles builds them, social_nav_sim.py:381-387) + 3 polygon walls   (configs[4])"""
    cases = []
    seed = 0
    for t, model in enumerate(SFMS):
        for kind, n, radius, warm in (("n10", 10, 2.8, 100), ("n50", 50, 6.0, 110), ("n50_walls_static", 50, 6.4, 110)):
            seed += 1
            rng = np.random.default_rng(130_000 + seed)
            walls = my_walls(rng) if kind == "n50_walls_static" else None
            cfg = crossing_config(rng, model, n, radius, walls=walls)
            if kind == "n50_walls_static":
                for i in range(3):
                    h = cfg["humans"][i]
                    ang = 2.1 * i + 1.35          # half way between the walls of my_walls (at 2.1 k + 0.3), clear of them
                    h["pos"] = [float(2.4 * math.cos(ang)), float(2.4 * math.sin(ang))]
                    h["goals"] = [list(h["pos"]), list(h["pos"])]
                    h["des_speed"] = 0.0
                    h["radius"] = 0.8
            sim = ns.sim.SocialNavSim(cfg, scenario="custom_config", parallelize_humans=True)
            mm = sim.motion_model_manager
            for k in range(warm):
                mm.update_humans(0, DT)
            before = mm_snapshot(mm)
            for k in range(20):
                mm.update_humans(0, DT)
            after = mm_snapshot(mm)
            cases.append(dict(kind=kind, model=model, type=t, n=n, dt=DT, n_substeps=20,
                              all_params_equal=bool(mm.all_equal_humans), respawn=False,
                              **{f"in_{k}": v for k, v in before.items()},
                              out_states=after["states"], out_goals=after["goals"]))
        # the reference's own parallel-traffic generator at 25 humans, respawns inside the 20-substep window
        seed += 1
        np.random.seed(1300 + seed)
        kw = {"insert_robot": False, "human_policy": model, "headless": True, "runge_kutta": False, "n_actors": 25,
              "traffic_length": 14, "traffic_height": 3}
        sim = ns.sim.SocialNavSim(kw, scenario="parallel_traffic", parallelize_humans=True)
        mm = sim.motion_model_manager
        for k in range(60):
            mm.update_humans(0, DT)
        for i in (2, 11, 19):
            mm.states[i, 0] = mm.goals[i, 0, 0] + 3.0 + 0.015 * (i + 1)
            mm.humans[i].set_state(mm.states[i, 0:8])
        before = mm_snapshot(mm)
        for k in range(20):
            mm.update_humans(0, DT)
        after = mm_snapshot(mm)
        cases.append(dict(kind="n25_traffic", model=model, type=t, n=25, dt=DT, n_substeps=20,
                          all_params_equal=bool(mm.all_equal_humans), respawn=bool(mm.parallel_traffic_humans_respawn),
                          respawn_bounds=[float(x) for x in mm.respawn_bounds],
                          **{f"in_{k}": v for k, v in before.items()},
                          out_states=after["states"], out_goals=after["goals"]))
        sim.parallel_traffic_humans_respawn = False
    print("g13_block_sizes:", len(cases), "cases ->", save_cases("g13_block_sizes", cases))


def gen_g16_policy():
    """The reference's own value-based policies deciding in the reference's own Gym: CADRL (cadrl.py:235-291) and SARL
    (multi_human_rl.py:12-88 with sarl.py's attention network), default policy.config, weights drawn from a seeded torch generator,
    the parallel branch (SocialNavGym.reset sets policy.parallelize, social_nav_gym.py:128: compute_rotated_states_and_reward + one
    model() call per action).  Recorded per decision: the full crowd rows the env holds (mm.states, goal lists), the robot's full state,
    the observation, what get_next_human_observable_states returned to the policy, the 81 action values and the arg-max."""
    import torch
    import crowd_nav.policy.cadrl as cadrl_mod
    import crowd_nav.policy.multi_human_rl as mh_mod
    from crowd_nav.policy.policy_factory import policy_factory
    from crowd_nav.utils.state import JointState

    pcfg = configparser.RawConfigParser()
    pcfg.read(os.path.join(_refharness.REFERENCE_ROOT, "crowd_nav", "configs", "policy.config"))
    cases = []
    for pname, human_policy, n, scenario in (("cadrl", "sfm_helbing", 5, "circle_crossing"), ("sarl", "hsfm_farina", 5, "circle_crossing"),
                                             ("sarl", "hsfm_new_guo", 10, "parallel_traffic")):
        env, cfg = make_env(human_policy, scenario, n, robot_visible=False)
        policy = policy_factory[pname]()
        policy.configure(pcfg)
        torch.manual_seed({"cadrl": 1601, "sarl": 1602}[pname] + n)
        with torch.no_grad():                       # an untrained network whose 81 values are spread out: weights a few times the default init
            for prm in policy.model.parameters():
                prm.copy_(torch.randn_like(prm) * (0.25 if prm.dim() > 1 else 0.1))
        policy.set_device(torch.device("cpu"))
        policy.set_phase("test")
        policy.set_env(env)
        env.robot.set_policy(policy)
        # the weights once per policy instance (float32, as torch holds them): a case of its own, the decisions name it by `wkey`
        wkey = f"{pname}_{human_policy}_{n}"
        sd = policy.model.state_dict()
        cases.append(dict(kind="weights", wkey=wkey, policy=pname, weights_keys=np.array(list(sd.keys())),
                          **{f"w_{i}": v.detach().cpu().numpy().astype(np.float32) for i, v in enumerate(sd.values())}))
        rec = {}
        orig_peek = env.motion_model_manager.__class__.get_next_human_observable_states
        orig_cav_c, orig_cav_m = cadrl_mod.compute_action_value, mh_mod.compute_action_value

        def cav(rewards, outs, dt, gamma, vpref):
            v = orig_cav_c(rewards, outs, dt, gamma, vpref)
            rec["values"], rec["rewards"], rec["net"] = np.array(v), np.array(rewards), np.array(outs)
            return v
        cadrl_mod.compute_action_value = mh_mod.compute_action_value = cav
        try:
            for test_case in range(4):
                ob, _ = env.reset(phase="test", test_case=test_case)
                mm = env.motion_model_manager

                def peek(self, dt, theta_and_omega_visible=False, _o=orig_peek):
                    r = _o(self, dt, theta_and_omega_visible)
                    rec["next"] = np.array(r)
                    return r
                mm.__class__.get_next_human_observable_states = peek
                for k in range(14):
                    rec.clear()
                    snap = mm_snapshot(mm)
                    robot_full = env.robot.get_full_state()
                    action = env.robot.act(ob)
                    if "values" not in rec:      # reach_destination: no decision was taken
                        break
                    A = policy.action_space_ndarray
                    cases.append(dict(kind="decision", wkey=wkey, policy=pname, model=human_policy, type=SFMS.index(human_policy), scenario=scenario, n=n,
                                      test_case=test_case, step=k, gamma=policy.gamma, dt=env.robot_time_step, substep=env.time_step, action_space=np.array(A),
                                      mm_states=snap["states"], mm_goals=snap["goals"], mm_params=snap["params"], mm_safety=snap["safety"],
                                      robot=np.array([robot_full.px, robot_full.py, robot_full.vx, robot_full.vy, robot_full.radius, robot_full.gx,
                                                      robot_full.gy, robot_full.v_pref, robot_full.theta]),
                                      obs=ob_to_array(ob), next_humans=rec["next"], rewards=rec["rewards"], net_outputs=rec["net"],
                                      action_values=rec["values"], chosen=int(np.argmax(rec["values"])), action=np.array([action.vx, action.vy])))
                    ob, reward, term, trunc, info = env.step(action)
                    if term or trunc:
                        break
        finally:
            cadrl_mod.compute_action_value, mh_mod.compute_action_value = orig_cav_c, orig_cav_m
            env.motion_model_manager.__class__.get_next_human_observable_states = orig_peek
    print("g16_policy:", sum(c["kind"] == "decision" for c in cases), "decisions ->", save_cases("g16_policy", cases))


def gen_g15_imitation_rk45():
    """The Gym seam with the ROBOT integrated by RK45: set_human_motion_model_as_robot_policy(model, runge_kutta=True)
    (social_nav_sim.py:862-873 -> motion_model_manager.py:552-563) + imitation_learning_step (social_nav_gym.py:252-274): every one of
    the 20 substeps of a Gym step is update_robot (solve_ivp RK45 over dt around compute_robot_forces, motion_model_manager.py:631-640,
    :661-687) followed by the crowd's Euler update_humans.  Six SFM / HSFM robot models, robot visible or not, three scenarios."""
    cases = []
    seed = 0
    human_models = ["sfm_helbing", "hsfm_farina", "sfm_guo", "hsfm_new_guo"]
    scenarios = ["circle_crossing", "parallel_traffic", "hybrid_scenario"]
    for rmodel in ("sfm_helbing", "sfm_guo", "hsfm_farina", "hsfm_guo", "hsfm_new", "hsfm_new_guo"):
        for robot_visible in (False, True):
            seed += 1
            rng = np.random.default_rng(150_000 + seed)
            hmodel = human_models[seed % len(human_models)]
            scen = scenarios[seed % len(scenarios)]
            env, _ = make_env(hmodel, scen, 5, robot_visible)
            env.set_human_motion_model_as_robot_policy(rmodel, True)
            if seed % 4 == 0:
                env.set_safety_space(0.1)
            phase = ("test", "val", "train")[seed % 3]
            test_case = int(rng.integers(0, 90))
            ob, info = env.reset(phase=phase, test_case=test_case)
            mm = env.motion_model_manager
            assert mm.robot_runge_kutta is True
            obs, rewards, terms, truncs, infos = [ob_to_array(ob)], [], [], [], []
            robots = [robot_row(env.robot)]
            mm_states, mm_goals = [mm.states.copy()], [mm.goals.copy()]
            for k in range(10):
                ob, r, term, trunc, info = env.imitation_learning_step()
                obs.append(ob_to_array(ob)); rewards.append(float(r)); terms.append(bool(term)); truncs.append(bool(trunc))
                infos.append(type(info[0]).__name__)
                robots.append(robot_row(env.robot))
                mm_states.append(mm.states.copy()); mm_goals.append(mm.goals.copy())
            cases.append(dict(robot_model=rmodel, model=hmodel, scenario=scen, robot_visible=robot_visible, phase=phase, test_case=test_case,
                              human_num=5, safety_space=float(env.safety_space), respawn=bool(mm.parallel_traffic_humans_respawn),
                              obs=np.array(obs), rewards=np.array(rewards), terminated=np.array(terms), truncated=np.array(truncs), infos=infos,
                              robots=np.array(robots), mm_states=np.array(mm_states), mm_goals=np.array(mm_goals), mm_safety=mm.safety_space.copy(),
                              human_safety=np.array([h.safety_space for h in mm.humans], dtype=float),
                              robot_params=env.robot.get_parameters(rmodel), global_time=float(env.global_time)))
            env.parallel_traffic_humans_respawn = False
    print("g15_imitation_rk45:", len(cases), "cases ->", save_cases("g15_imitation_rk45", cases))


def gen_g17_unicycle():
    """the unicycle robot (RobotAgent.step / compute_position with ActionRot, robot_agent.py:119-136) inside the Gym's
    substep loop (social_nav_gym.py:240-245: robot.step(action, dt); update_humans(t, dt) -- time_step_factor times, the SAME (v, r) every
    substep, so the robot turns by r per SUBSTEP).  The reference's own step() cannot take an ActionRot (social_nav_sim.py:973 reads
    robot.theta, which no agent has), so the loop of :240-245 is run here directly on the reference's objects; every substep's robot pose and
    crowd rows are recorded.  Also recorded, with the attribute the reference reads provided (robot.theta := robot.yaw): what
    collision_detection_and_reaching_goal + compute_reward_and_infos return for that action (the Gym head's swept test, :949-1029)."""
    cases = []
    seed = 0
    for model, scen, hn, visible in (("sfm_helbing", "circle_crossing", 5, True), ("sfm_helbing", "circle_crossing", 5, False),
                                     ("hsfm_farina", "circle_crossing", 10, True), ("hsfm_new_guo", "parallel_traffic", 10, True),
                                     ("hsfm_farina", "hybrid_scenario", 25, True), ("hsfm_farina", "hybrid_scenario", 25, False),
                                     ("hsfm_new_guo", "circle_crossing", 25, True), ("sfm_guo", "parallel_traffic", 25, True),
                                     ("hsfm_new_moussaid", "circle_crossing", 10, True)):
        for rep in range(2):
            seed += 1
            rng = np.random.default_rng(170_000 + seed)
            env, _ = make_env(model, scen, hn, visible, False, "unicycle")
            if seed % 3 == 0:
                env.set_safety_space(0.1)
            test_case = int(rng.integers(0, 90))
            env.reset(phase="test", test_case=test_case)
            mm = env.motion_model_manager
            robot = env.robot
            if rep == 1:       # a heading that is not a multiple of pi / 2, a robot inside the crowd's reach
                robot.yaw = float(rng.uniform(-np.pi, np.pi))
                robot.position = np.array(robot.position, dtype=np.float64) * 0.5
            if model.endswith("moussaid"):   # everybody at rest: sign(theta_ij ~ 0) follows float64 noise in the reference -> record a moving crowd
                for _ in range(env.time_step_factor):
                    robot.step(ns.action.ActionRot(0.6, 0.03), env.time_step)
                    mm.update_humans(env.global_time, env.time_step)
                    env.global_time += env.time_step
            rec = dict(model=model, scenario=scen, human_num=hn, robot_visible=visible, safety_space=float(env.safety_space), test_case=test_case, rep=rep,
                       global_time0=float(env.global_time), respawn=bool(mm.parallel_traffic_humans_respawn), all_params_equal=bool(mm.all_equal_humans),
                       dt=float(env.time_step), n_substeps=int(env.time_step_factor), T=float(env.robot_time_step),
                       mm_safety=mm.safety_space.copy(), params=mm.params.copy(),
                       respawn_bounds=np.array(getattr(env, "respawn_bounds", None) or (0.0, 0.0), dtype=np.float64),
                       robot_radius=float(robot.radius), robot_safety_space=float(getattr(robot, "safety_space", 0.0)), sfm_type=int(mm.sfm_type), robot_goal=np.array(robot.get_goal_position(), dtype=np.float64))
            actions, robots, states, goals, heads = [], [], [], [], []
            robot0 = robot.get_safe_state().copy()
            states0, goals0 = mm.states.copy(), mm.goals.copy()
            n_steps = 3
            for k in range(n_steps):
                v = float(rng.uniform(0.3, 1.0))
                r = float(rng.uniform(-0.08, 0.08)) if k else float(rng.choice([-1.0, 1.0]) * rng.uniform(0.02, 0.08))
                action = ns.action.ActionRot(v, r)
                actions.append([v, r])
                # the Gym head for this action (social_nav_gym.py:229-233) with the attribute :973 reads
                robot.theta = robot.yaw
                col, dmin, reach = env.collision_detection_and_reaching_goal(action, env.robot_time_step)
                reward, term, trunc, info = env.compute_reward_and_infos(col, dmin, reach, env.global_time, env.robot_time_step)
                del robot.theta
                heads.append([float(col), float(dmin), float(reach), float(reward), float(term), float(trunc)])
                rec.setdefault("infos", []).append(type(info).__name__)
                rr, ss, gg = [], [], []
                for _ in range(env.time_step_factor):        # social_nav_gym.py:240-245
                    robot.step(action, env.time_step)
                    mm.update_humans(env.global_time, env.time_step)
                    env.global_time += env.time_step
                    rr.append(robot.get_safe_state().copy()); ss.append(mm.states.copy()); gg.append(mm.goals.copy())
                robots.append(rr); states.append(ss); goals.append(gg)
            rec.update(robot0=robot0, states0=states0, goals0=goals0, actions=np.array(actions), heads=np.array(heads),
                       robots=np.array(robots), states=np.array(states), goals=np.array(goals))
            cases.append(rec)
            env.parallel_traffic_humans_respawn = False
    # RobotAgent.step called on its own (no crowd): 200 substeps of one action -- the yaw wraps through 2 pi (python's % keeps it in [0, 2 pi))
    env, _ = make_env("sfm_helbing", "circle_crossing", 5, False, False, "unicycle")
    env.reset(phase="test", test_case=3)
    for v, r, yaw0 in ((0.8, 0.11, 1.5), (0.5, -0.07, -3.0), (1.0, 0.0, 0.3), (0.0, 0.2, 6.0)):
        robot = env.robot
        robot.position = np.array([0.3, -0.2]); robot.yaw = yaw0; robot.linear_velocity = np.zeros(2)
        robot0 = robot.get_safe_state().copy()
        rr = []
        for _ in range(200):
            robot.step(ns.action.ActionRot(v, r), 0.0125)
            rr.append(robot.get_safe_state().copy())
        cases.append(dict(model="none", robot_only=True, robot0=robot0, actions=np.array([[v, r]]), robots=np.array([rr]), dt=0.0125, n_substeps=200))
    print("g17_unicycle:", len(cases), "cases ->", save_cases("g17_unicycle", cases))


GROUPS = dict(g1_direct=gen_g1_direct, g1_episode=gen_g1_episode, g2_block=gen_g2_block, g3_gym=gen_g3_gym,
              g4_peek=gen_g4_peek, g5_reward=gen_g5_reward, g6_generators=gen_g6_generators,
              g7_respawn=gen_g7_respawn, g8_lookahead=gen_g8_lookahead, g9_laser=gen_g9_laser,
              g10_social_momentum=gen_g10_social_momentum, g11_imitation=gen_g11_imitation, g12_rk45=gen_g12_rk45, g14_rk45_more=gen_g14_rk45_more,
              g13_block_sizes=gen_g13_block_sizes, g15_imitation_rk45=gen_g15_imitation_rk45, g16_policy=gen_g16_policy, g17_unicycle=gen_g17_unicycle)

if __name__ == "__main__":
    todo = sys.argv[1:] or list(GROUPS)
    for g in todo:
        GROUPS[g]()
