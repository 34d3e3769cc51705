"""MotionModelManager -- host mirror of the reference class for the accelerated path.

Reference: social_gym/src/motion_model_manager.py.  Same constructor, attributes (``states``, ``goals``,
``params``, ``obstacles``, ``safety_space``, ``sfm_type``, ``headed``, ``all_equal_humans``,
``parallel_traffic_humans_respawn``, ``respawn_bounds`` ...) and methods on the path CrowdNav uses:

  update_humans(t, dt, post_update=True)                      :354-422   (Euler SFM / HSFM and ORCA)
  get_human_states / set_human_states                         :285-352
  get_next_human_observable_states(dt, theta_and_omega_visible) :691-709  (peek)
  set_safety_space(safety_space)                              :147-170
  update_goals / rewind_goals / bound_velocity                :52-70

plus ``update_humans_block(dt, n_substeps, action)``: the substep loop of SocialNavGym.step fused in
one kernel launch.  The arithmetic runs in libcrowdstep.so on the GPU (float32); the numpy arrays kept
here are float64 mirrors refreshed after every call, and each ``HumanAgent`` holds views into
``self.states`` exactly as in the reference (agent.py:260-266).  Out of scope on this path, as in the
Gym env: RK45 integration, social momentum, the robot's own SFM / ORCA policies (they raise).
"""
from __future__ import annotations

import numpy as np

from ... import _lib
from ...batched import CrowdWorlds, HUMAN_MODELS, ORCA_DEFAULTS as _ORCA, SFMS
from .agent import Agent
from .forces_parallel import mirror_goal_rotation
from .utils import PRECISION, bound_angle

N_GENERAL_STATES = 8
N_HEADED_STATES = 6
N_NOT_HEADED_STATES = 4
ORCA_DEFAULTS = [10, 10, 5, 5]  # neighbor_dist, max_neighbors, time_horizon, time_horizon_obstacles


class MotionModelManager:
    def __init__(self, motion_model_title: str, consider_robot: bool, runge_kutta: bool, humans: list, robot, walls: list,
                 parallelize=False):
        self.consider_robot = consider_robot
        self.runge_kutta = runge_kutta
        self.update_targets = True
        self.humans = humans
        self.robot = robot
        self.walls = walls
        self.parallel = True if motion_model_title in SFMS else parallelize  # the device path IS the array path
        self.orca = False
        self.sm = False
        self.sf = False
        self.parallel_traffic_humans_respawn = False
        self.respawn_bounds = None
        self._cw = None
        self.set_human_motion_model(motion_model_title)
        self.robot_motion_model_title = None

    # ------------------------------------------------------------------ small helpers (:52-70)
    def bound_velocity(self, velocity, desired_speed):
        nrm = np.linalg.norm(velocity)
        return (velocity / nrm) * desired_speed if nrm > desired_speed else velocity

    def rewind_goals(self, agent: Agent, goal: list):
        if goal not in agent.goals:
            agent.set_goals([goal])
        elif agent.goals and agent.goals[0] != goal:
            while agent.goals[0] != goal:
                agent.goals.append(agent.goals.pop(0))

    def update_goals(self, agent: Agent):
        if agent.goals and np.linalg.norm(np.asarray(agent.goals[0]) - agent.position) < agent.radius:
            agent.goals.append(agent.goals.pop(0))

    def headed_agent_update_linear_velocity(self, agent: Agent):
        agent.compute_rotational_matrix()
        agent.linear_velocity = np.matmul(agent.rotational_matrix, agent.body_velocity)

    # ------------------------------------------------------------------ model set-up (:222-283)
    def set_human_motion_model(self, motion_model_title: str):
        self.motion_model_title = motion_model_title
        if motion_model_title not in HUMAN_MODELS and motion_model_title != "social_momentum":
            if motion_model_title == "sfm_roboticsupo":
                raise NotImplementedError(f"'{motion_model_title}' is not reachable from SocialNavGym and not on the accelerated path")
            raise Exception(f"The human motion model '{motion_model_title}' does not exist")
        n = len(self.humans)
        rows = n + int(self.consider_robot)
        self.safety_space = np.zeros(rows, PRECISION)
        max_n_goals = int(np.max([len(h.goals) for h in self.humans]))
        self.goals = np.full((n, max_n_goals, 2), np.nan, PRECISION)
        for i, h in enumerate(self.humans):
            for j, goal in enumerate(h.goals):
                self.goals[i, j] = np.array(goal, PRECISION)
        if motion_model_title == "social_momentum":  # (:247-251) the step itself runs in csrc/social_momentum.hip
            self.sm, self.orca, self.headed, self.include_mass = True, False, False, False
            self.sfm_type = _lib.CS_SOCIAL_MOMENTUM
            self.params = None
            self.obstacles = None  # the model ignores walls (:395-404)
            self.n_actions = 20
            self.actions_angles = [((2 * np.pi) / self.n_actions) * i for i in range(self.n_actions)]
            for h in self.humans:
                h.action_set = [np.array([np.cos(a), np.sin(a)], dtype=PRECISION) * h.desired_speed for a in self.actions_angles]
            self.states = np.array([h.get_safe_state() for h in self.humans], PRECISION)
            if self.consider_robot:
                self.states = np.append(self.states, [self.robot.get_safe_state()], axis=0)
            self.all_equal_humans = True
        elif motion_model_title == "orca":
            self.orca, self.headed, self.include_mass = True, False, False
            self.sfm_type = _lib.CS_ORCA
            self.params = None
            self.obstacles = None
            # sim.addObstacle(list(wall.vertices)) for every wall, then processObstacles (:244-246)
            from ...rvo2 import process_obstacles
            self._orca_vertices = process_obstacles([np.asarray(w.vertices) for w in self.walls]) if len(self.walls) > 0 else None
            self._orca_margin = 0.01  # agent.radius + 0.01 at addAgent (:241)
            for i in range(n):  # update_goals_orca at creation (:242)
                self.update_goals(self.humans[i])
            self._sync_goal_array_from_lists()
            self.states = np.array([h.get_safe_state() for h in self.humans], PRECISION)
            if self.consider_robot:
                self.states = np.append(self.states, [self.robot.get_safe_state()], axis=0)
            self._refresh_orca_pref(range(n))
            self.all_equal_humans = True
        else:
            self.orca = False
            self.headed = motion_model_title.startswith("hsfm")
            self.include_mass = True
            self.type = SFMS.index(motion_model_title) % 3
            self.sfm_type = SFMS.index(motion_model_title)
            for h in self.humans:
                h.set_parameters(motion_model_title)
            self.states = np.array([h.get_safe_state() for h in self.humans], PRECISION)
            if self.consider_robot:
                self.states = np.append(self.states, [self.robot.get_safe_state()], axis=0)
            self.params = np.array([h.get_parameters(motion_model_title) for h in self.humans], PRECISION)
            if len(self.walls) > 0:
                smax = int(np.max([len(o.segments) for o in self.walls]))
                self.obstacles = np.full((len(self.walls), smax, 2, 2), np.nan, PRECISION)
                for i, o in enumerate(self.walls):
                    for j, seg in o.segments.items():
                        self.obstacles[i, j, 0] = np.array(seg[0], PRECISION)
                        self.obstacles[i, j, 1] = np.array(seg[1], PRECISION)
            else:
                self.obstacles = None
            # the reference's pairwise check `break`s only the inner loop (:278-283): the flag ends up
            # as the verdict of the LAST compared pair of the last row that compared anything
            self.all_equal_humans = True
            for i in range(n):
                for j in range(i + 1, n):
                    self.all_equal_humans = self.check_pair_agents_social_force_parameters(self.humans[i], self.humans[j])
                    if not self.all_equal_humans:
                        break
        self._attach_views()
        self._cw = None

    def check_pair_agents_social_force_parameters(self, a1: Agent, a2: Agent):
        if a1.radius != a2.radius or a1.mass != a2.mass:
            return False
        m = self.motion_model_title
        if m in ("sfm_helbing", "hsfm_farina", "hsfm_new"):
            names = ["Ai", "Bi", "k1", "k2"]
        elif m.endswith("guo"):
            names = ["Ai", "Bi", "Ci", "Di", "k1", "k2"]
        elif m.endswith("moussaid"):
            names = ["agent_lambda", "gamma", "Ei", "ns1", "ns", "k1", "k2"]
        else:
            raise NotImplementedError(f"The {m} model is not implemented")
        return all(getattr(a1, k) == getattr(a2, k) for k in names)

    # ------------------------------------------------------------------ device plumbing
    def _attach_views(self):
        for i, h in enumerate(self.humans):
            h.set_state(self.states[i, 0:8])

    def _sync_goal_array_from_lists(self):
        n, G = self.goals.shape[0], self.goals.shape[1]
        for i, h in enumerate(self.humans):
            k = len(h.goals)
            if k > G:
                raise ValueError("a human has more goals than the manager was built for")
            self.goals[i, :k] = np.array(h.goals, PRECISION).reshape(k, 2)
            self.goals[i, k:] = np.nan

    def _sync_goal_lists_from_array(self):
        for i, h in enumerate(self.humans):
            row = self.goals[i]
            valid = ~np.isnan(row).any(axis=1)
            h.goals = [[float(g[0]), float(g[1])] for g in row[valid]]

    def _refresh_orca_pref(self, idx):
        """setAgentPrefVelocity of update_goals_orca (:128-132): unit direction, or the raw difference
        when the goal is closer than desired_speed."""
        for i in idx:
            h = self.humans[i]
            d = np.array(h.goals[0], dtype=PRECISION) - h.position
            nrm = np.linalg.norm(d)
            self.states[i, 5:7] = d / nrm if nrm > h.desired_speed else d
            self.states[i, 10:12] = h.goals[0]

    def _device(self, respawn: bool = None) -> CrowdWorlds:
        """(Re)build the resident world from the host mirrors: they are the reference's public, mutable
        arrays, so every call starts from what the caller sees."""
        if self.consider_robot:
            rb = self.robot.get_safe_state()
            if self.orca:  # the simulator's robot agent keeps what set_state_orca last gave it
                rb = self.states[-1]
            self.states[-1] = rb
        if self.sm:  # the filter reads every entity's own safety_space attribute (social_momentum.py:22)
            for i, h in enumerate(self.humans):
                self.safety_space[i] = h.safety_space
            if self.consider_robot:
                self.safety_space[-1] = self.robot.safety_space
        margin = self.safety_space + (self._orca_margin if self.orca else 0.0)
        robot_rows = self.robot.get_safe_state() if (self.robot is not None and len(self.robot.goals) > 0) else None
        if respawn is None:
            respawn = self.parallel_traffic_humans_respawn
        bounds = self.respawn_bounds if respawn else None
        verts = getattr(self, "_orca_vertices", None) if self.orca else None
        # The world stays RESIDENT between calls (round 6): the host mirrors are the reference's public, mutable arrays, so rows, goal lists and
        # the robot row are uploaded at every call -- into the buffers of the previous call when nothing about their shapes or the kernel build
        # changed; parameters, margins and walls only when their content did.  (Rebuilding the batch per call -- seven hipMalloc / hipFree pairs --
        # was 130 of the 230 us a W = 1 step spent outside its launch: tools/facade_latency.py.)
        sig = (self.states.shape, self.goals.shape, int(self.sfm_type), bool(self.all_equal_humans), bool(self.consider_robot), robot_rows is not None,
               None if self.obstacles is None else np.shape(self.obstacles), None if verts is None else np.shape(verts), np.shape(self.params))
        cw = getattr(self, "_cw", None)
        if cw is not None and getattr(self, "_cw_sig", None) == sig:
            cw.set_states(self.states)
            cw.set_goals(self.goals)
            if robot_rows is not None:
                cw.set_robot(robot_rows)
            held = self._cw_held
            if not np.array_equal(margin, held["margin"]):
                cw.set_safety(margin); held["margin"] = np.array(margin, copy=True)
            if self.params is not None and not np.array_equal(self.params, held["params"]):
                cw.d_params.upload(np.asarray(self.params, dtype=np.float32).reshape(cw.d_params.shape), cw.stream); held["params"] = np.array(self.params, copy=True)
            if self.obstacles is not None and not np.array_equal(self.obstacles, held["obstacles"], equal_nan=True):
                cw.d_obstacles.upload(np.asarray(self.obstacles, dtype=np.float32), cw.stream); held["obstacles"] = np.array(self.obstacles, copy=True)
            if verts is not None and not np.array_equal(verts, held["verts"]):
                cw = None                                  # (RVO2 vertex records changed: rebuild)
            else:
                cw.respawn_bounds = bounds
                cw.unicycle = False
                return cw
        self._cw = CrowdWorlds(self.states, self.goals, self.params, margin, self.obstacles, type=self.sfm_type,
                               all_params_equal=self.all_equal_humans, robot_row=self.consider_robot,
                               robot=robot_rows, respawn_bounds=bounds, orca_vertices=verts)
        self._cw_sig = sig
        self._cw_held = dict(margin=np.array(margin, copy=True), params=None if self.params is None else np.array(self.params, copy=True),
                             obstacles=None if self.obstacles is None else np.array(self.obstacles, copy=True),
                             verts=None if verts is None else np.array(verts, copy=True))
        return self._cw

    def _readback(self, cw: CrowdWorlds, robot_moved=False):
        new = cw.get_states()[0].astype(PRECISION)
        n = len(self.humans)
        if self.orca:
            self.states[:n, [0, 1, 3, 4, 5, 6, 10, 11]] = new[:n][:, [0, 1, 3, 4, 5, 6, 10, 11]]
            if self.consider_robot:
                self.states[n, [0, 1, 3, 4]] = new[n, [0, 1, 3, 4]]
        else:
            self.states[:n, 0:8] = new[:n, 0:8]
            self.states[:n, 10:12] = new[:n, 10:12]
            if self.consider_robot:
                self.states[n] = new[n]
        mirror_goal_rotation(self.goals, cw.get_goals()[0])
        self._sync_goal_lists_from_array()
        if robot_moved:
            rb = cw.get_robot()[0].astype(PRECISION)
            self.robot.position = rb[0:2].copy()
            self.robot.yaw = float(rb[2])
            self.robot.linear_velocity = rb[3:5].copy()

    # ------------------------------------------------------------------ the hot path
    def update_humans(self, t: float, dt: float, post_update=True):
        """One substep of every human (:354-422): Euler SFM / HSFM, or ORCA; parallel-traffic respawn when
        ``post_update``."""
        if self.runge_kutta and not (self.orca or self.sm):   # RK45 of the SFM / HSFM crowd (:374-384); ORCA is Euler only
            cw = self._device(respawn=bool(post_update and self.parallel_traffic_humans_respawn))   # respawn behind the solve (:405-422)
            if not hasattr(self, "_desired_force"):   # agent.desired_force of the single-agent force functions (forces.py:12-16)
                self._desired_force = np.zeros((len(self.humans), 2), dtype=PRECISION)
            self.rk45_nfev = int(cw.update_humans_rk45(dt, desired_force=self._desired_force[None])[0])
            self._readback(cw)
            self._desired_force = cw.d_rk_memory.download(cw.stream)[0].astype(PRECISION)
            for i, h in enumerate(self.humans):
                h.desired_force = self._desired_force[i]
            return
        cw = self._device(respawn=bool(post_update and self.parallel_traffic_humans_respawn))
        cw.step(dt, 1, None)
        self._readback(cw)
        if self.orca and self.consider_robot:  # set_state_orca(robot) after doStep (:389, :111-114)
            n = len(self.humans)
            self.states[n, 0:2] = self.robot.position
            self.states[n, 3:5] = self.robot.linear_velocity

    def complete_rk45_simulation(self, t: float, dt: float, final_time: float):
        """ONE adaptive RK45 solve over (t, t + final_time) with the solution sampled at np.arange(t, final_time, dt) through the
        solver's dense output (:461-498) -> human_states [len(times), n, 6 | 4] (x, y, yaw, BVx, BVy, Omega | x, y, Vx, Vy).  The humans
        are left as the last right-hand-side evaluation leaves them (the state at t + final_time), like the reference's."""
        if not self.runge_kutta or self.orca or self.sm:
            raise ValueError("complete_rk45_simulation integrates an SFM / HSFM crowd created with runge_kutta=True")
        times = np.arange(t, final_time, dt, dtype=PRECISION)
        cw = self._device(respawn=False)
        if not hasattr(self, "_desired_force"):
            self._desired_force = np.zeros((len(self.humans), 2), dtype=PRECISION)
        out, nfev = cw.complete_rk45_simulation(dt, final_time, len(times), desired_force=self._desired_force[None])
        self.rk45_nfev = int(nfev[0])
        self._readback(cw)
        self._desired_force = cw.d_rk_memory.download(cw.stream)[0].astype(PRECISION)
        for i, h in enumerate(self.humans):
            h.desired_force = self._desired_force[i]
        return out[0].astype(PRECISION)

    def update_humans_block(self, dt: float, n_substeps: int, action=None, unicycle=False):
        """``n_substeps`` x { robot.step(action, dt) ; update_humans(t, dt) } fused in one launch -- the loop of
        SocialNavGym.step (social_nav_gym.py:240-245).  ``action``: (vx, vy) or (v, r) for a unicycle robot."""
        if self.runge_kutta and not (self.orca or self.sm):
            raise NotImplementedError("the fused Gym block is Euler (SocialNavGym never sets runge_kutta, social_nav_gym.py:141-143); "
                                      "call update_humans(t, dt) for RK45")
        cw = self._device()
        cw.unicycle = bool(unicycle)
        act = None if action is None else np.asarray(action, dtype=np.float32).reshape(1, 2)
        cw.step(dt, n_substeps, act)
        self._readback(cw, robot_moved=action is not None)

    # ------------------------------------------------------------------ state access (:285-352)
    def get_human_states(self, include_goal=True, headed=False):
        n = len(self.humans)
        cols = N_GENERAL_STATES if include_goal else (N_HEADED_STATES if headed else N_NOT_HEADED_STATES)
        state = np.empty([n, cols], dtype=PRECISION)
        for i, h in enumerate(self.humans):
            v = h.body_velocity if headed else h.linear_velocity
            if include_goal:
                state[i] = [h.position[0], h.position[1], h.yaw, v[0], v[1], h.angular_velocity, h.goals[0][0], h.goals[0][1]]
            elif headed:
                state[i] = [h.position[0], h.position[1], h.yaw, v[0], v[1], h.angular_velocity]
            else:
                state[i] = [h.position[0], h.position[1], v[0], v[1]]
        return state

    def set_human_states(self, state, just_visual=False):
        """State rows [x, y, yaw, Vx|BVx, Vy|BVy, Omega, Gx, Gy] (:314-345)."""
        n = len(self.humans)
        if just_visual:
            for i in range(n):
                self.states[i, 0:3] = state[i, 0:3]
            return
        for i in range(n):
            s = self.states[i]
            s[0:3] = state[i, 0:3]
            if self.headed:
                s[5:7] = state[i, 3:5]
                c, sn = np.cos(s[2]), np.sin(s[2])
                s[3:5] = [c * s[5] - sn * s[6], sn * s[5] + c * s[6]]
            else:
                s[3:5] = state[i, 3:5]
            s[7] = state[i, 5]
            goal = [float(state[i, 6]), float(state[i, 7])]
            self.rewind_goals(self.humans[i], goal)
            s[10:12] = state[i, 6:8]
        self._sync_goal_array_from_lists()
        if self.orca:  # set_state_orca(i): goal check + preferred velocity from the restored position (:105-110)
            for i in range(n):
                self.update_goals(self.humans[i])
            self._sync_goal_array_from_lists()
            self._refresh_orca_pref(range(n))

    def get_next_human_observable_states(self, dt: float, theta_and_omega_visible=False):
        """Next human states after ONE Euler step of size dt, without committing it (:691-709).
        [N, 4] = px, py, vx, vy  or  [N, 8] = x, y, yaw, Vx, Vy, Omega, Gx, Gy."""
        cw = self._device()
        nxt = cw.peek(dt)[0].astype(PRECISION)
        if self.headed:  # set_human_states(saved) recomputes the linear velocity from the body velocity (:324,333)
            n = len(self.humans)
            c, s = np.cos(self.states[:n, 2]), np.sin(self.states[:n, 2])
            bx, by = self.states[:n, 5].copy(), self.states[:n, 6].copy()
            self.states[:n, 3] = c * bx - s * by
            self.states[:n, 4] = s * bx + c * by
        if not (self.orca or self.sm):
            # set_human_states(saved) also rewrites the goal columns of every state row from the saved state, whose goal is the head of
            # the human's goal LIST (:285-297, :333: `*state[i,6:8]`).  Where the row and the list had disagreed (parallel traffic after a
            # respawn: the row keeps the clamped y, golden g4_peek cases 2 and 6) the row follows the list from the first peek on -- and
            # the next peek's desired force with it (4e-5 on a velocity in case 6).
            n = len(self.humans)
            self.states[:n, 10:12] = np.array([h.goals[0] for h in self.humans], dtype=PRECISION).reshape(n, 2)
        if self.orca:
            self._refresh_orca_pref(range(len(self.humans)))
        return nxt if theta_and_omega_visible else nxt[:, [0, 1, 3, 4]]

    # ------------------------------------------------------------------ safety space (:147-170)
    def set_safety_space(self, safety_space: float):
        if self.motion_model_title is not None:
            if "sfm" in self.motion_model_title:
                for i, h in enumerate(self.humans):
                    h.safety_space = 0.01 + safety_space
                    self.safety_space[i] = 0.01 + safety_space
            elif self.motion_model_title == "orca":
                # sim.setAgentRadius(i, radius + 0.01 + safety_space) for humans AND the robot agent
                self.safety_space[:] = safety_space
            else:
                raise NotImplementedError(f"Model {self.motion_model_title} is not implemented for humans")
        if self.robot_motion_model_title is not None:   # robot safety space (:160-170)
            if "sfm" in self.robot_motion_model_title:
                self.robot.safety_space = 0.01 + safety_space
                if self.parallel and self.consider_robot:
                    self.safety_space[len(self.humans)] = 0.01 + safety_space
            elif self.robot_motion_model_title == "orca":
                # robot_sim.setAgentRadius(i, radius + 0.01 + safety_space) for the humans AND the robot of the robot's simulator
                self._robot_sim_margin = 0.01 + safety_space
            else:
                raise NotImplementedError(f"Model {self.motion_model_title} is not implemented for robot")

    # ------------------------------------------------------------------ the robot under a human motion model (:552-653)
    def set_robot_motion_model(self, motion_model_title: str, runge_kutta: bool):
        """The robot follows one of the human motion models ("sfm_helbing" ... "hsfm_new_moussaid", "orca"); ``runge_kutta``: the
        SFM / HSFM robot is integrated by scipy's RK45 (cs_robot_model_rk45) instead of Euler (:631-640)."""
        if motion_model_title not in SFMS + ["orca"]:
            if motion_model_title == "sfm_roboticsupo":
                raise NotImplementedError("sfm_roboticsupo is outside this build (DESIGN.md §9)")
            raise Exception(f"The robot motion model '{motion_model_title}' does not exist")
        if runge_kutta and motion_model_title == "orca":
            raise ValueError("ORCA is Euler only")
        self.robot_runge_kutta = bool(runge_kutta)
        self.robot_motion_model_title = motion_model_title
        self.robot_orca = motion_model_title == "orca"
        self.robot.orca = self.robot_orca
        self.robot_headed = motion_model_title.startswith("hsfm")
        self.robot.headed = self.robot_headed
        self.robot_include_mass = not self.robot_orca
        self._robot_sim_margin = 0.01      # addAgent(..., radius + 0.01, ...) of the robot's own simulator (:585-587)
        if not hasattr(self.robot, "desired_force"):
            self.robot.desired_force = np.zeros(2, dtype=PRECISION)
        if not self.robot_orca:
            self.robot.set_parameters(motion_model_title)

    def _device_with_robot_model(self) -> CrowdWorlds:
        if self.robot_motion_model_title is None:
            raise AttributeError("set_robot_motion_model has not been called")
        cw = self._device()
        n = len(self.humans)
        if self.robot_orca:
            hm = np.full(cw.rows, self._robot_sim_margin, dtype=np.float32)
            verts = getattr(self, "_orca_vertices", None)
            if verts is None and self.walls:
                from ...rvo2 import process_obstacles
                verts = process_obstacles([list(w.vertices) for w in self.walls])
            cw.set_robot_model("orca", None, self._robot_sim_margin, hm, orca_vertices=verts)
        else:
            hm = np.zeros(cw.rows, dtype=np.float32)
            hm[:n] = [h.safety_space for h in self.humans]
            cw.set_robot_model(self.robot_motion_model_title, self.robot.get_parameters(self.robot_motion_model_title),
                               self.robot.safety_space, hm)
            cw.d_robot_memory.upload(np.asarray(self.robot.desired_force, dtype=np.float32).reshape(1, 2), cw.stream)
        return cw

    def _readback_robot(self, cw: CrowdWorlds):
        rb = cw.get_robot()[0].astype(PRECISION)
        self.robot.position = rb[0:2].copy()
        self.robot.linear_velocity = rb[3:5].copy()
        if not self.robot_orca:
            self.robot.yaw = float(rb[2])
            self.robot.body_velocity = rb[5:7].copy()
            self.robot.angular_velocity = float(rb[7])
            self.robot.desired_force = cw.d_robot_memory.download(cw.stream)[0].astype(PRECISION)

    def update_robot(self, t, dt, just_velocities=False):
        """One substep of the robot under its motion model (:615-653)."""
        cw = self._device_with_robot_model()
        if getattr(self, "robot_runge_kutta", False) and not self.robot_orca:
            if just_velocities:   # the reference's own error (:631)
                raise ValueError("Runge-kutta integration cannot be used if robot and environment sampling times are different")
            self.robot_rk45_nfev = int(cw.robot_model_rk45(dt)[0])
        else:
            cw.robot_model_step(dt, just_velocities=just_velocities)
        self._readback_robot(cw)

    def imitation_block(self, dt: float, n_substeps: int):
        """``n_substeps`` x { update_robot(t, dt) ; update_humans(t, dt) } without leaving the device -- the loop of
        SocialNavGym.imitation_learning_step (social_nav_gym.py:259-263)."""
        cw = self._device_with_robot_model()
        if getattr(self, "robot_runge_kutta", False) and not self.robot_orca:
            # the robot under RK45 (:631-640): one solve over dt per substep with the humans standing, then the crowd's Euler substep --
            # the reference's strict alternation, 2 x n_substeps launches, one read-back
            nfev = 0
            for _ in range(int(n_substeps)):
                nfev += int(cw.robot_model_rk45(dt)[0])
                cw.step(dt, 1, None)
            self.robot_rk45_nfev = nfev
        else:
            cw.imitation_block(dt, n_substeps)
        self._readback(cw)
        self._readback_robot(cw)
        if self.orca and self.consider_robot:  # set_state_orca(robot) after the last doStep (:389)
            n = len(self.humans)
            self.states[n, 0:2] = self.robot.position
            self.states[n, 3:5] = self.robot.linear_velocity

    # ------------------------------------------------------------------ robot state access (:520-550)
    def get_robot_state(self, include_goal=True, headed=False):
        """[x, y, yaw, (B)Vx, (B)Vy, Omega, Gx, Gy] with the goal; without it [x, y, yaw, BVx, BVy, Omega] (headed) or
        [x, y, Vx, Vy]."""
        r = self.robot
        v = r.body_velocity if headed else r.linear_velocity
        if include_goal:
            return np.array([r.position[0], r.position[1], r.yaw, v[0], v[1], r.angular_velocity, r.goals[0][0], r.goals[0][1]], dtype=PRECISION)
        if headed:
            return np.array([r.position[0], r.position[1], r.yaw, v[0], v[1], r.angular_velocity], dtype=PRECISION)
        return np.array([r.position[0], r.position[1], v[0], v[1]], dtype=PRECISION)

    def set_robot_state(self, state):
        """Inverse of get_robot_state(include_goal=True, headed=robot.headed); rewinds the goal list to (Gx, Gy)."""
        r = self.robot
        r.position[0], r.position[1], r.yaw = state[0], state[1], state[2]
        if not r.headed:
            r.linear_velocity[0], r.linear_velocity[1] = state[3], state[4]
        else:
            r.body_velocity[0], r.body_velocity[1] = state[3], state[4]
        r.angular_velocity = state[5]
        self.rewind_goals(r, [state[6], state[7]])
        if self.consider_robot and self.orca:  # set_state_orca(robot): the crowd simulator's copy of the robot
            n = len(self.humans)
            self.states[n, 0:2] = r.position
            self.states[n, 3:5] = r.linear_velocity

    def update_robot_pose(self, dt: float):
        self.robot.position += self.robot.linear_velocity * dt
        self.robot.yaw += self.robot.angular_velocity * dt
