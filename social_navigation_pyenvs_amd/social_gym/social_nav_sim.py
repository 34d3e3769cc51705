"""SocialNavSim -- headless simulator shell behind SocialNavGym (reference: social_gym/social_nav_sim.py).

Kept from the reference (same names and semantics): the three scenario generators with the legacy
``np.random`` draw order (:200-431), ``reset_sim`` (:97-198), ``set_time_step`` / ``set_robot_time_step``
(:82-95), ``collision_detection_and_reaching_goal`` (:949-984), ``compute_reward_and_infos`` (:986-1029),
``onestep_lookahead`` (:1031-1049), ``transform_human_states`` / constant-velocity propagation
(:935-947, 1051-1066) and a headless ``run_k_steps`` (:670-714).  Everything PyGame (window, sprites,
rewind, plots, manual driving) is out of scope: rendering is bypassed on this path.

Rejection loops are bounded (``MAX_PLACEMENT_TRIES``): the reference's ``while True`` loops never end
for infeasible requests (e.g. 5 actors in the static-obstacle scenario).
"""
from __future__ import annotations

import math

import numpy as np

from ..crowd_nav.utils.state import ObservableState, ObservableStateHeaded
from .src.agent import HumanAgent, RobotAgent
from .src.info import Collision, Danger, Nothing, ReachGoal, Timeout
from .src.motion_model_manager import N_GENERAL_STATES, MotionModelManager  # noqa: F401
from .src.obstacle import Obstacle
from .src.utils import PRECISION, bound_angle, is_multiple, point_to_segment_dist

MAX_FPS = 60
SAMPLING_TIME = 1 / MAX_FPS
ROBOT_SAMPLING_TIME = 1 / 4
REAL_SIZE = 15
MOTION_MODELS = ["sfm_roboticsupo", "sfm_helbing", "sfm_guo", "sfm_moussaid", "hsfm_farina", "hsfm_guo",
                 "hsfm_moussaid", "hsfm_new", "hsfm_new_guo", "hsfm_new_moussaid", "orca"]
MAX_PLACEMENT_TRIES = 100000


class SocialNavSim:
    def __init__(self, config_data, scenario="custom_config", parallelize_robot=False, parallelize_humans=False):
        self.real_size = REAL_SIZE
        self.display_to_real_ratio = 1.0
        self.walls = []
        self.humans = []
        self.mode = scenario
        self.parallelize_robot = parallelize_robot
        self.parallelize_humans = parallelize_humans
        self.sampling_time = SAMPLING_TIME
        self.robot_sampling_time = ROBOT_SAMPLING_TIME
        if scenario == "custom_config":
            self.config_data = config_data
        elif scenario == "circular_crossing":
            self.config_data = self.generate_circular_crossing_setting(**config_data)
        elif scenario == "parallel_traffic":
            self.config_data = self.generate_parallel_traffic_scenario(**config_data)
        elif scenario == "circular_crossing_with_static_obstacles":
            self.config_data = self.generate_circular_crossing_with_static_obstacles(**config_data)
        else:
            raise Exception(f"Scenario '{scenario}' does not exist")
        self.reset_sim(restart_gui=True)

    # ------------------------------------------------------------------ time steps (:82-95)
    def set_time_step(self, time_step: float):
        self.sampling_time = time_step
        self.robot_env_same_timestep = (self.sampling_time == self.robot_sampling_time)

    def set_robot_time_step(self, time_step: float):
        if not is_multiple(time_step, self.sampling_time):
            raise ValueError("Robot time step must be a multiple of the environment sampling time")
        self.robot_sampling_time = time_step
        self.robot.time_step = time_step
        if self.robot.policy is not None:
            self.robot.policy.time_step = time_step
        self.robot_env_same_timestep = (self.sampling_time == self.robot_sampling_time)

    # ------------------------------------------------------------------ reset (:97-198)
    def reset_sim(self, restart_gui=False, reset_robot=True):
        cfg = self.config_data
        self.humans.clear()
        self.walls = []
        self.headless = cfg.get("headless", False)
        self.motion_model = cfg.get("motion_model", "sfm_helbing")
        self.runge_kutta = cfg.get("runge_kutta", False)
        self.grid = cfg.get("grid", True)
        self.robot_visible = cfg.get("robot_visible", False)
        if reset_robot:
            if "robot" in cfg:
                r = cfg["robot"]
                self.robot = RobotAgent(self, list(r["pos"]), r["yaw"], r["radius"], [list(g) for g in r["goals"]])
                self.insert_robot = True
            else:
                self.robot = RobotAgent(self)
                self.insert_robot = False
        self.updated = True
        for wall in cfg["walls"]:
            self.walls.append(Obstacle(self, wall))
        for key in cfg["humans"]:
            h = cfg["humans"][key]
            self.humans.append(HumanAgent(self, key, self.motion_model, list(h["pos"]), h["yaw"], [list(g) for g in h["goals"]],
                                          h.get("color", (0, 0, 0)), h.get("radius", 0.3), h.get("mass", 75),
                                          h.get("des_speed", 0.9), h.get("group_id", -1)))
        if self.motion_model == "orca":
            self.parallelize_humans = False
        # a robot that follows a human motion model keeps it across resets (:174-179)
        previous = getattr(self, "motion_model_manager", None)
        robot_model = previous.robot_motion_model_title if previous is not None else None
        robot_rk = previous.robot_runge_kutta if robot_model is not None else False
        if robot_model is None and getattr(self, "_pending_robot_model", None) is not None:
            robot_model, robot_rk = self._pending_robot_model   # asked for before the first world existed
        self.motion_model_manager = MotionModelManager(self.motion_model, self.robot_visible, self.runge_kutta, self.humans,
                                                       self.robot, self.walls, parallelize=self.parallelize_humans)
        self.robot_controlled = False
        if robot_model is not None:
            self.motion_model_manager.set_robot_motion_model(robot_model, robot_rk)
            self.robot_crowdnav_policy = False
            self.robot_controlled = True
        # sticky, as in the reference (:183-185): once a parallel-traffic scenario was generated on this
        # object, every later reset keeps the respawn rule switched on
        if getattr(self, "parallel_traffic_humans_respawn", False):
            self.motion_model_manager.parallel_traffic_humans_respawn = True
            self.motion_model_manager.respawn_bounds = self.respawn_bounds
        self.robot_env_same_timestep = (self.sampling_time == self.robot_sampling_time)
        self.n_updates = 0
        self.sim_t = 0.0

    # ------------------------------------------------------------------ scenario generators (:200-431)
    @staticmethod
    def _attributes(n_actors, randomize):
        speeds, radii = [], []
        for _ in range(n_actors):  # one speed draw then one radius draw per human (:217-220)
            if randomize:
                speeds.append(np.random.uniform(0.5, 1.5))
                radii.append(np.random.uniform(0.3, 0.5))
            else:
                speeds.append(1.0)
                radii.append(0.3)
        return speeds, radii

    def generate_circular_crossing_setting(self, **kwargs):
        insert_robot = kwargs.get("insert_robot", False)
        model = kwargs.get("human_policy", "sfm_guo")
        headless = kwargs.get("headless", False)
        runge_kutta = kwargs.get("runge_kutta", False)
        robot_visible = kwargs.get("robot_visible", False)
        robot_r = kwargs.get("robot_radius", 0.3)
        radius = kwargs.get("circle_radius", 7)
        n_actors = kwargs.get("n_actors", 10)
        rand = kwargs.get("randomize_human_positions", False)
        speeds, radii = self._attributes(n_actors, kwargs.get("randomize_human_attributes", False))
        cx = cy = 0.0
        humans = {}
        robot = None
        if insert_robot:
            robot = {"pos": [cx, cy - radius], "yaw": math.pi / 2, "radius": robot_r, "goals": [[cx, cy + radius], [cx, cy - radius]]}
        if not rand:
            slots = n_actors + (1 if insert_robot else 0)
            step = (2 * math.pi) / slots
            for i in range(n_actors):
                k = i + 1 if insert_robot else i
                off = -(math.pi / 2) if insert_robot else 0.0
                px, py = radius * math.cos(off + step * k), radius * math.sin(off + step * k)
                yaw = bound_angle((math.pi / 2) + step * k) if insert_robot else bound_angle(-math.pi + step * k)
                humans[i] = {"pos": [cx + px, cy + py], "yaw": yaw, "goals": [[cx - px, cy - py], [cx + px, cy + py]],
                             "des_speed": speeds[i], "radius": radii[i]}
        else:
            placed = []
            robot_pos = np.array([cx, cy - radius], dtype=PRECISION)
            robot_goal = np.array([cx, cy + radius], dtype=PRECISION)
            for i in range(n_actors):
                for _ in range(MAX_PLACEMENT_TRIES):
                    angle = np.random.random() * np.pi * 2
                    noise = np.array([(np.random.random() - 0.5) * speeds[i], (np.random.random() - 0.5) * speeds[i]], dtype=PRECISION)
                    pos = np.array([cx + radius * np.cos(angle) + noise[0], cy + radius * np.sin(angle) + noise[1]], dtype=PRECISION)
                    collide = False
                    for j, other in enumerate(placed):
                        min_dist = radii[i] + radii[j] + 0.2
                        o_pos = np.array(other, dtype=PRECISION)
                        o_goal = np.array([-other[0] + 2 * cx, -other[1] + 2 * cx], dtype=PRECISION)  # (:280: centre x twice)
                        if np.linalg.norm(pos - o_pos) < min_dist or np.linalg.norm(pos - o_goal) < min_dist:
                            collide = True
                            break
                    if insert_robot and (np.linalg.norm(pos - robot_pos) < radii[i] + robot_r + 0.2
                                         or np.linalg.norm(pos - robot_goal) < radii[i] + robot_r + 0.2):
                        collide = True
                    if not collide:
                        placed.append([pos[0], pos[1]])
                        humans[i] = {"pos": [pos[0], pos[1]], "yaw": bound_angle(math.pi + angle),
                                     "goals": [[cx * 2 - pos[0], cy * 2 - pos[1]], [pos[0], pos[1]]],
                                     "des_speed": speeds[i], "radius": radii[i]}
                        break
                else:
                    raise RuntimeError("circular crossing: could not place all humans (circle too small for n_actors)")
        data = {"motion_model": model, "headless": headless, "runge_kutta": runge_kutta,
                "robot_visible": robot_visible if insert_robot else False, "grid": True, "walls": [], "humans": humans}
        if insert_robot:
            data["robot"] = robot
        self.config_data = data
        return data

    def generate_parallel_traffic_scenario(self, **kwargs):
        insert_robot = kwargs.get("insert_robot", False)
        human_policy = kwargs.get("human_policy", "sfm_guo")
        headless = kwargs.get("headless", False)
        runge_kutta = kwargs.get("runge_kutta", False)
        robot_visible = kwargs.get("robot_visible", False)
        robot_radius = kwargs.get("robot_radius", 0.3)
        L = kwargs.get("traffic_length", 14)
        H = kwargs.get("traffic_height", 3)
        n_actors = kwargs.get("n_actors", 10)
        robot = None
        if insert_robot:
            robot = {"pos": [-(L / 2) + 1, 0], "yaw": 0.0, "radius": robot_radius, "goals": [[(L / 2) - 1, 0], [-(L / 2) + 1, 0]]}
        speeds, radii = self._attributes(n_actors, kwargs.get("randomize_human_attributes", False))
        if sum(math.pi * (r ** 2) for r in radii) > L * H * 0.4:
            raise ValueError("Number of humans specified is too big for desided traffic height and length")
        humans, placed = {}, []
        for i in range(n_actors):
            for _ in range(MAX_PLACEMENT_TRIES):
                a, b = -(L / 2) + radii[i], L / 2 - radii[i]
                pos = np.array([(b - a) * np.random.random() + a, (np.random.random() - 0.5) * H], dtype=PRECISION)
                collide = False
                for j, other in enumerate(placed):
                    if np.linalg.norm(pos - other) - radii[i] - radii[j] - 0.1 < 0:
                        collide = True
                        break
                if insert_robot and np.linalg.norm(pos - np.array(robot["pos"], PRECISION)) - radii[i] - robot["radius"] - 0.1 < 0:
                    collide = True
                if not collide:
                    placed.append(pos)
                    humans[i] = {"pos": [pos[0], pos[1]], "yaw": bound_angle(-math.pi), "goals": [[-(L / 2) - 3, pos[1]]],
                                 "des_speed": speeds[i], "radius": radii[i]}
                    break
            else:
                raise RuntimeError("parallel traffic: could not place all humans")
        data = {"motion_model": human_policy, "headless": headless, "runge_kutta": runge_kutta,
                "robot_visible": robot_visible if insert_robot else False, "grid": True, "walls": [], "humans": humans}
        if insert_robot:
            data["robot"] = robot
        self.parallel_traffic_humans_respawn = True
        self.respawn_bounds = ((L / 2), (H / 2))
        self.config_data = data
        return data

    def generate_circular_crossing_with_static_obstacles(self, **kwargs):
        insert_robot = kwargs.get("insert_robot", False)
        model = kwargs.get("human_policy", "sfm_guo")
        headless = kwargs.get("headless", False)
        runge_kutta = kwargs.get("runge_kutta", False)
        robot_visible = kwargs.get("robot_visible", False)
        robot_r = kwargs.get("robot_radius", 0.3)
        radius = kwargs.get("circle_radius", 7)
        n_actors = kwargs.get("n_actors", 10)
        assert radius > 5, "Radius must be greater than 5 for this scenario"
        inner = radius - 3
        cx = cy = 0.0
        speeds, radii = [], []
        for i in range(n_actors):  # the three "obstacles" are immobile humans with a drawn radius (:381-387)
            if i < 3:
                speeds.append(0.0)
                radii.append(1 + (np.random.random() - 1) * 0.4)
            else:
                speeds.append(1.0)
                radii.append(0.3)
        humans, placed = {}, []
        robot_pos = np.array([cx, cy - radius], dtype=PRECISION)
        robot_goal = np.array([cx, cy + radius], dtype=PRECISION)
        sector = np.pi / int(n_actors / 2)
        for i in range(n_actors):
            for _ in range(MAX_PLACEMENT_TRIES):
                if i < 3:
                    angle = sector * (-0.5 + 2 * i + (np.random.random() - 0.5) * 0.5)
                    noise = np.array([(np.random.random() - 0.5) * 0.1, (np.random.random() - 0.5) * 0.1], dtype=PRECISION)
                    ring = inner
                else:
                    angle = sector * (0.5 + 2 * i + (np.random.random() - 0.5) * 0.5)
                    noise = np.array([(np.random.random() - 0.5) * 0.7, (np.random.random() - 0.5) * 0.7], dtype=PRECISION)
                    ring = radius
                pos = np.array([cx + ring * np.cos(angle) + noise[0], cy + ring * np.sin(angle) + noise[1]], dtype=PRECISION)
                collide = False
                for j, other in enumerate(placed):
                    min_dist = radii[i] + radii[j] + 0.2
                    o_pos = np.array(other, dtype=PRECISION)
                    o_goal = o_pos if j < 3 else np.array([-other[0] + 2 * cx, -other[1] + 2 * cx], dtype=PRECISION)
                    if np.linalg.norm(pos - o_pos) < min_dist or np.linalg.norm(pos - o_goal) < min_dist:
                        collide = True
                        break
                if np.linalg.norm(pos - robot_pos) < radii[i] + robot_r + 0.2 or np.linalg.norm(pos - robot_goal) < radii[i] + robot_r + 0.2:
                    collide = True
                if not collide:
                    placed.append([pos[0], pos[1]])
                    goals = [[pos[0], pos[1]], [pos[0], pos[1]]] if i < 3 else [[cx * 2 - pos[0], cy * 2 - pos[1]], [pos[0], pos[1]]]
                    humans[i] = {"pos": [pos[0], pos[1]], "yaw": bound_angle(math.pi + angle), "goals": goals,
                                 "des_speed": speeds[i], "radius": radii[i]}
                    break
            else:
                raise RuntimeError("static obstacles: could not place all humans (n_actors < 6 puts two obstacles on one spot)")
        data = {"motion_model": model, "headless": headless, "runge_kutta": runge_kutta,
                "robot_visible": robot_visible if insert_robot else False, "grid": True, "walls": [], "humans": humans}
        if insert_robot:
            data["robot"] = {"pos": [cx, cy - radius], "yaw": math.pi / 2, "radius": robot_r, "goals": [[cx, cy + radius], [cx, cy - radius]]}
        self.config_data = data
        return data

    # ------------------------------------------------------------------ headless stepping
    def update(self):
        """One simulator update of sampling_time (humans only; the robot is moved by the caller)."""
        self.motion_model_manager.update_humans(self.sim_t, self.sampling_time)
        self.sim_t += self.sampling_time
        self.n_updates += 1
        self.updated = True

    def run_k_steps(self, steps, quit=True, additional_info=False, stop_when_collision_or_goal=False, save_states_time_step=None):
        """Headless rollout of the humans (:670-714, without the robot controller): returns the human states
        [steps, N, 8] (x, y, yaw, Vx|BVx, Vy|BVy, Omega, Gx, Gy)."""
        mm = self.motion_model_manager
        out = np.empty((steps, len(self.humans), N_GENERAL_STATES), dtype=PRECISION)
        for k in range(steps):
            out[k] = mm.get_human_states(include_goal=True, headed=mm.headed)
            self.update()
        return out

    # ------------------------------------------------------------------ CrowdNav hooks (:935-1066)
    def transform_human_states(self, state, theta_and_omega_visible=False):
        out = []
        for i, hs in enumerate(state):
            if theta_and_omega_visible:
                out.append(ObservableStateHeaded(hs[0], hs[1], hs[3], hs[4], self.humans[i].radius, hs[2], hs[5]))
            else:
                out.append(ObservableState(hs[0], hs[1], hs[2], hs[3], self.humans[i].radius))
        return out

    def set_human_motion_model_as_robot_policy(self, policy_name, runge_kutta):
        """The robot moves towards its goal with a human motion model (:862-873); used by imitation learning."""
        if getattr(self, "motion_model_manager", None) is None:   # no world yet (the Gym builds it at the first reset)
            if runge_kutta and policy_name == "orca":
                raise NotImplementedError                          # the reference's own (:579)
            self._pending_robot_model = (policy_name, bool(runge_kutta))
            return
        self.motion_model_manager.set_robot_motion_model(policy_name, runge_kutta)

    def collision_detection_and_reaching_goal(self, action, time_step):
        """Swept robot-human test over `time_step` with the humans' current velocities (:949-984)."""
        dmin = float("inf")
        collision = False
        if isinstance(action, np.ndarray):
            robot_velocity = action
        elif self.robot.kinematics == "holonomic":
            robot_velocity = np.array([action.vx, action.vy])
        else:
            # the reference reads self.robot.theta here (:973), an attribute RobotAgent only gets (as None) in
            # configure(): the unicycle branch cannot run there; the robot's yaw is what it means
            robot_velocity = np.array([action.v * np.cos(action.r + self.robot.yaw), action.v * np.sin(action.r + self.robot.yaw)])
        for human in self.humans:
            d = human.position - self.robot.position
            e = d + (human.linear_velocity - robot_velocity) * time_step
            closest = point_to_segment_dist(d[0], d[1], e[0], e[1], 0, 0) - human.radius - self.robot.radius
            if closest < 0:
                collision = True
                break
            elif closest < dmin:
                dmin = closest
        if isinstance(action, np.ndarray):
            end_position = self.robot.position + action * time_step
        else:
            end_position = self.robot.compute_position(action, time_step)
        reaching_goal = np.linalg.norm(end_position - self.robot.get_goal_position()) < self.robot.radius
        return collision, dmin, reaching_goal

    def compute_reward_and_infos(self, collision, dmin, reaching_goal, current_time, time_step):
        if current_time >= self.time_limit - 1:
            return 0, False, True, Timeout()
        if collision:
            return self.collision_penalty, True, False, Collision()
        if reaching_goal:
            return self.success_reward, True, False, ReachGoal()
        if dmin < self.discomfort_dist:
            return (dmin - self.discomfort_dist) * self.discomfort_penalty_factor * time_step, False, False, Danger(dmin)
        return 0, False, False, Nothing()

    def onestep_lookahead(self, action, time_step=None):
        if time_step is None:
            time_step = self.robot_sampling_time
        collision, dmin, reaching_goal = self.collision_detection_and_reaching_goal(action, time_step)
        reward, _, _, _ = self.compute_reward_and_infos(collision, dmin, reaching_goal, self.sim_t, time_step)
        if not self.updated:
            ob = self.last_observation.copy()
        else:
            pol = self.robot.policy
            if pol.query_env:
                headed = pol.with_theta_and_omega_visible
                nxt = self.motion_model_manager.get_next_human_observable_states(time_step, theta_and_omega_visible=headed)
                ob = self.transform_human_states(nxt, theta_and_omega_visible=headed)
            else:
                ob = self.propagate_humans_state_with_constant_velocity_model(time_step)
            self.last_observation = ob.copy()
        self.updated = False
        return ob, reward

    def propagate_humans_state_with_constant_velocity_model(self, time_step):
        out = []
        headed = self.robot.policy.with_theta_and_omega_visible
        for h in self.humans:
            p = h.position + h.linear_velocity * time_step
            if headed:
                out.append(ObservableStateHeaded(*p, *h.linear_velocity, h.radius, h.yaw + h.angular_velocity * time_step, h.angular_velocity))
            else:
                out.append(ObservableState(*p, *h.linear_velocity, h.radius))
        return out

    # ------------------------------------------------------------------ rendering: bypassed
    def render_sim(self):
        raise NotImplementedError("rendering is bypassed on the MI355X crowd-step path")

    run_live = run_and_plot_trajectories_humans = render_sim
