"""SocialNavGym -- the drop-in Gym boundary (reference: social_gym/social_nav_gym.py:14-277).

``configure(config)``, ``set_robot(robot)``, ``set_safety_space(s)``, ``reset(phase, test_case)``,
``step(action)``, ``render()`` and the attributes CrowdNav reads (``global_time, time_limit, time_step,
robot_time_step, case_size, case_counter, humans, robot, motion_model, states, updated,
motion_model_manager``) keep the reference's names, argument meaning, return shapes and error
behaviour (``AttributeError`` without a robot, ``ValueError`` for a robot time step that is not a
multiple of the time step, ``NotImplementedError`` for an unknown human policy).

What differs underneath: the ``time_step_factor`` substeps of ``step()`` (robot move + ``update_humans``
+ respawn, :240-245) are ONE fused kernel launch on the MI355X; PyGame is gone (``render`` is a no-op as
with the reference's ``HEADLESS = True``); gymnasium is optional (the class subclasses ``gym.Env`` when
the package is importable).  ``BatchedSocialNavGym`` below runs W such worlds in lock-step.
"""
from __future__ import annotations

import logging

import numpy as np

from .. import _lib
from ..batched import CrowdWorlds, HUMAN_MODELS as _MODELS
from .social_nav_sim import SocialNavSim
from .src.info import INFO_BY_CODE, Collision, Danger, Nothing, ReachGoal, Timeout  # noqa: F401
from .src.utils import is_multiple

try:  # optional: register / subclass when gymnasium exists (it is not part of this image)
    import gymnasium as gym

    _EnvBase = gym.Env
except Exception:  # pragma: no cover
    gym = None

    class _EnvBase:  # minimal stand-in so isinstance checks on the API shape still work
        pass

HEADLESS = True
PARALLELIZE_ROBOT = True
PARALLELIZE_HUMANS = True
HUMAN_MODELS = list(_MODELS)


class SocialNavGym(_EnvBase, SocialNavSim):
    def __init__(self):
        # the reference builds a throw-away 10-human world here (:25); an empty shell is enough
        self.config_data = {"humans": {}, "walls": [], "headless": True}
        self.humans = []
        self.walls = []
        self.mode = "circular_crossing"
        self.parallelize_robot = PARALLELIZE_ROBOT
        self.parallelize_humans = PARALLELIZE_HUMANS
        self.sampling_time = 1 / 60
        self.robot_sampling_time = 1 / 4
        self.real_size = 15
        self.display_to_real_ratio = 1.0
        self.updated = True
        self.sim_t = 0.0
        self.time_limit = None
        self.time_step = None
        self.robot = None
        self.global_time = None
        self.human_times = None
        self.success_reward = None
        self.collision_penalty = None
        self.discomfort_dist = None
        self.discomfort_penalty_factor = None
        self.config = None
        self.case_capacity = None
        self.case_size = None
        self.case_counter = None
        self.randomize_attributes = None
        self.train_val_sim = None
        self.test_sim = None
        self.square_width = None
        self.circle_radius = None
        self.human_num = None
        self.safety_space = 0
        self.states = None
        self.action_values = None
        self.attention_weights = None
        self.action_space = gym.spaces.Discrete(1) if gym is not None else None
        self.observation_space = gym.spaces.Discrete(1) if gym is not None else None

    # ------------------------------------------------------------------ configuration (:59-98)
    def configure(self, config):
        self.config = config
        self.time_limit = config.getint("env", "time_limit")
        self.time_step = config.getfloat("env", "time_step")
        self.robot_time_step = config.getfloat("env", "robot_time_step")
        if not is_multiple(self.robot_time_step, self.time_step):
            raise ValueError("Robot time step must be a multiple of time step")
        self.time_step_factor = int(self.robot_time_step / self.time_step)
        self.randomize_attributes = config.getboolean("env", "randomize_attributes")
        self.success_reward = config.getfloat("reward", "success_reward")
        self.collision_penalty = config.getfloat("reward", "collision_penalty")
        self.discomfort_dist = config.getfloat("reward", "discomfort_dist")
        self.discomfort_penalty_factor = config.getfloat("reward", "discomfort_penalty_factor")
        self.human_policy = config.get("humans", "policy")
        self.robot_radius = config.getfloat("robot", "radius")
        if self.human_policy not in HUMAN_MODELS:
            raise NotImplementedError
        u32 = np.iinfo(np.uint32).max
        self.case_capacity = {"train": u32 - 2000, "val": 1000, "test": 1000}
        self.case_size = {"train": u32 - 2000, "val": config.getint("env", "val_size"), "test": config.getint("env", "test_size")}
        self.train_val_sim = config.get("sim", "train_val_sim")
        self.test_sim = config.get("sim", "test_sim")
        self.traffic_height = config.getfloat("sim", "traffic_height")
        self.traffic_length = config.getfloat("sim", "traffic_length")
        self.circle_radius = config.getfloat("sim", "circle_radius")
        self.human_num = config.getint("sim", "human_num")
        self.case_counter = {"train": 0, "test": 0, "val": 0}
        logging.info("human number: {}".format(self.human_num))
        logging.info("Human policy: {}".format(self.human_policy))

    def set_safety_space(self, safety_space: float):
        self.safety_space = safety_space

    def set_robot(self, robot):
        self.robot = robot
        if PARALLELIZE_ROBOT:
            self.robot.parallelize = True

    def compute_humans_observable_state(self):
        if self.robot.sensor == "coordinates":
            headed = self.robot.policy.with_theta_and_omega_visible
            return [h.get_observable_state(visible_theta_and_omega=True) if headed else h.get_observable_state() for h in self.humans]
        raise NotImplementedError

    def check_actual_collisions_and_goal(self):
        dmin, collision = 10000.0, False
        for h in self.humans:
            d = np.linalg.norm(h.position - self.robot.position) - h.radius - self.robot.radius
            dmin = d if d < dmin else dmin
            if dmin <= 0:
                collision = True
        reaching_goal = np.linalg.norm(self.robot.position - self.robot.get_goal_position()) < self.robot.radius
        return collision, dmin, reaching_goal

    # ------------------------------------------------------------------ reset (:120-225)
    def _generate(self, scenario, human_num):
        kw = dict(insert_robot=True, human_policy=self.human_policy, headless=HEADLESS, runge_kutta=False,
                  robot_visible=self.robot.visible, robot_radius=self.robot_radius, n_actors=human_num,
                  randomize_human_attributes=self.randomize_attributes)
        if scenario == "circle_crossing":
            self.generate_circular_crossing_setting(circle_radius=self.circle_radius, randomize_human_positions=True, **kw)
        elif scenario == "parallel_traffic":
            self.generate_parallel_traffic_scenario(traffic_length=self.traffic_length, traffic_height=self.traffic_height, **kw)
        elif scenario == "circular_crossing_with_static_obstacles":
            self.generate_circular_crossing_with_static_obstacles(circle_radius=self.circle_radius, randomize_human_positions=True, **kw)
        # any other name: the reference falls through its elif chain and keeps the previous config_data

    def reset(self, phase="test", test_case=None):
        if self.robot is None:
            raise AttributeError("robot has to be set!")
        assert phase in ["train", "val", "test"]
        if test_case is not None:
            self.case_counter[phase] = test_case
        if self.robot.parallelize:
            self.robot.policy.parallelize = True
        self.global_time = 0
        multi = self.robot.policy.multiagent_training
        self.human_times = [0] * (self.human_num if (phase == "test" or multi) else 1)
        if self.config.get("humans", "policy") == "trajnet":
            raise NotImplementedError
        counter_offset = {"train": self.case_capacity["val"] + self.case_capacity["test"], "val": 0, "test": self.case_capacity["val"]}
        if self.case_counter[phase] >= 0:
            seed = counter_offset[phase] + self.case_counter[phase]
            np.random.seed(seed)
            if phase in ["train", "val"]:
                human_num = self.human_num if multi else 1
                scenario = self.train_val_sim
            else:
                human_num = self.human_num
                scenario = self.test_sim
            if scenario == "hybrid_scenario":
                scenario = np.random.choice(["circle_crossing", "parallel_traffic"])
                np.random.seed(seed)  # the choice draw is discarded: the generator restarts the stream (:156-157)
            self._generate(str(scenario), human_num)
            self.case_counter[phase] = (self.case_counter[phase] + 1) % self.case_size[phase]
        else:
            assert phase == "test"
            if self.case_counter[phase] == -1:  # the reference's debugging case (:199-209)
                self.human_num = 3
                humans = {0: {"pos": [0, -6], "yaw": np.pi / 2, "goals": [[0, 5], [0, -6]]},
                          1: {"pos": [-5, -5], "yaw": np.pi / 2, "goals": [[-5, 5], [-5, -5]]},
                          2: {"pos": [5, -5], "yaw": np.pi / 2, "goals": [[5, 5], [5, -5]]}}
                robot = {"pos": [7.5, 7.5], "yaw": 0.0, "radius": 0.25, "goals": [[7.5, 7.5]]}
                self.config_data = {"headless": False, "motion_model": "sfm_helbing", "runge_kutta": False, "insert_robot": True,
                                    "grid": True, "humans": humans, "walls": [], "robot": robot}
            else:
                raise NotImplementedError
        self.robot.set_radius_and_update_graphics(self.robot_radius)
        r = self.config_data["robot"]
        self.robot.set(r["pos"][0], r["pos"][1], r["goals"][0][0], r["goals"][0][1], 0, 0, r["yaw"], w=0)
        self.reset_sim(reset_robot=False)
        if self.safety_space > 0:
            self.motion_model_manager.set_safety_space(self.safety_space)
        self.set_time_step(self.time_step)
        self.set_robot_time_step(self.robot_time_step)
        self.states = list()
        if hasattr(self.robot.policy, "action_values"):
            self.action_values = list()
        if hasattr(self.robot.policy, "get_attention_weights"):
            self.attention_weights = list()
        return self.compute_humans_observable_state(), {0: Nothing()}

    # ------------------------------------------------------------------ step (:227-250)
    def step(self, action):
        collision, dmin, reaching_goal = self.collision_detection_and_reaching_goal(action, self.robot_time_step)
        reward, terminated, truncated, info = self.compute_reward_and_infos(collision, dmin, reaching_goal, self.global_time,
                                                                            self.robot_time_step)
        self.states.append([self.robot.get_full_state(), [h.get_full_state() for h in self.humans]])
        if hasattr(self.robot.policy, "action_values"):
            self.action_values.append(self.robot.policy.action_values)
        if hasattr(self.robot.policy, "get_attention_weights"):
            self.attention_weights.append(self.robot.policy.get_attention_weights())
        # time_step_factor x { robot.step(action, dt) ; update_humans(t, dt) } -> one fused launch
        self.robot.check_validity(action)
        unicycle = self.robot.kinematics != "holonomic"
        act = (action.v, action.r) if unicycle else (action.vx, action.vy)
        self.motion_model_manager.update_humans_block(self.time_step, self.time_step_factor, act, unicycle=unicycle)
        for _ in range(self.time_step_factor):
            self.global_time += self.time_step
        self.updated = True
        return self.compute_humans_observable_state(), reward, terminated, truncated, {0: info}

    def imitation_learning_step(self):
        """One environment step with the robot following a human motion model (:252-274): reward and info come from the
        ACTUAL state at the end of the update."""
        self.states.append([self.robot.get_full_state(), [h.get_full_state() for h in self.humans]])
        # time_step_factor x { update_robot(t, dt) ; update_humans(t, dt) } on the device, one read-back
        self.motion_model_manager.imitation_block(self.time_step, self.time_step_factor)
        for _ in range(self.time_step_factor):
            self.global_time += self.time_step
        ob = self.compute_humans_observable_state()
        collision, dmin, reaching_goal = self.check_actual_collisions_and_goal()
        reward, terminated, truncated, info = self.compute_reward_and_infos(collision, dmin, reaching_goal, self.global_time,
                                                                            self.robot_time_step)
        self.updated = True
        return ob, reward, terminated, truncated, {0: info}

    def render(self):
        return None


class BatchedSocialNavGym:
    """W independent SocialNavGym worlds advanced in lock-step on one GPU.

    Same episode logic as ``SocialNavGym.step`` -- reward / termination from the swept collision test on the
    CURRENT state (cs_collision_reward), then ``time_step_factor`` fused substeps (cs_step) -- with arrays
    instead of object lists: observations ``[W, N, 5]`` (px, py, vx, vy, radius) or ``[W, N, 7]`` (+ theta,
    omega), rewards ``[W]``, terminated / truncated ``[W]`` bool, ``info_code [W]`` (index into
    ``social_gym.src.info.INFO_BY_CODE``).  Worlds are generated by the exact reference generators from
    seeds ``offset[phase] + case`` (one per world) through a scratch ``SocialNavGym``; every world needs the
    same human count and goal-slot count (a hybrid batch pads traffic goals with NaN).
    """

    def __init__(self, config, n_worlds: int, robot_visible=False, robot_radius=None, headed_obs=False):
        self.W = int(n_worlds)
        self.headed_obs = bool(headed_obs)
        self._proto = SocialNavGym()
        self._proto.configure(config)
        self.config = config
        self.time_step = self._proto.time_step
        self.robot_time_step = self._proto.robot_time_step
        self.time_step_factor = self._proto.time_step_factor
        self.time_limit = self._proto.time_limit
        self.robot_visible = bool(robot_visible)
        self.reward_cfg = (float(self.time_limit), self._proto.success_reward, self._proto.collision_penalty,
                           self._proto.discomfort_dist, self._proto.discomfort_penalty_factor)
        self.global_time = None
        self.cw = None

    def reset(self, phase="test", first_case=0, safety_space=0.0, device=False):
        """Generate W worlds from the seeds ``offset[phase] + (first_case + w) % case_size``.  ``device=False`` runs the
        host generators world by world (W serial rejection-sampling loops); ``device=True`` runs them in one launch of
        ``cs_generate_worlds`` (one GPU lane per world, same legacy random stream, same rows)."""
        if device:
            return self._reset_on_device(phase, first_case, safety_space)
        from .src.agent import RobotAgent
        from types import SimpleNamespace

        proto = self._proto
        rows_s, rows_g, rows_p, rows_r, respawn = [], [], [], [], []
        model = proto.human_policy
        for w in range(self.W):
            robot = RobotAgent(proto)
            robot.visible, robot.sensor, robot.kinematics = self.robot_visible, "coordinates", "holonomic"
            robot.policy = SimpleNamespace(multiagent_training=True, with_theta_and_omega_visible=self.headed_obs, kinematics="holonomic")
            proto.parallel_traffic_humans_respawn = False  # one fresh env per world: no sticky respawn flag
            proto.set_robot(robot)
            proto.reset(phase=phase, test_case=first_case + w)
            mm = proto.motion_model_manager
            rows_s.append(mm.states.copy())
            rows_g.append(mm.goals.copy())
            rows_p.append(None if mm.params is None else mm.params.copy())
            rows_r.append(robot.get_safe_state())
            respawn.append(1 if mm.parallel_traffic_humans_respawn else 0)
            bounds = mm.respawn_bounds if mm.parallel_traffic_humans_respawn else None
            if bounds is not None:
                self._bounds = bounds
        gmax = max(g.shape[1] for g in rows_g)
        n = rows_g[0].shape[0]
        goals = np.full((self.W, n, gmax, 2), np.nan)
        for w, g in enumerate(rows_g):
            goals[w, :, :g.shape[1]] = g
        S = np.stack(rows_s)
        P = None if rows_p[0] is None else np.stack(rows_p)
        self.n = n
        self.radius = S[:, :n, 8].astype(np.float32)
        margin = np.zeros((self.W, S.shape[1]))
        if model == "orca":
            margin += 0.01 + safety_space
        elif safety_space > 0:
            margin[:, :n] = 0.01 + safety_space
        any_respawn = any(respawn)
        self.cw = CrowdWorlds(S, goals, P, margin, None, type=model, all_params_equal=True, robot_row=self.robot_visible,
                              robot=np.stack(rows_r), respawn_bounds=self._bounds if any_respawn else None,
                              respawn_worlds=np.array(respawn, np.int32) if any_respawn else None)
        self.global_time = np.zeros(self.W, np.float32)
        return self.observe()

    def _reset_on_device(self, phase, first_case, safety_space):
        from .. import generators as gen
        from .. import scenarios as sc
        from .src.agent import RobotAgent

        proto = self._proto
        assert phase in ["train", "val", "test"]
        scenario = proto.train_val_sim if phase in ("train", "val") else proto.test_sim
        if scenario not in gen.SCENARIOS:
            raise NotImplementedError(f"no device generator for scenario {scenario!r}")
        model = proto.human_policy
        n = proto.human_num
        robot = RobotAgent(proto)
        robot.set_radius_and_update_graphics(proto.robot_radius)
        traffic = scenario in ("parallel_traffic", "hybrid_scenario")
        G = 1 if scenario == "parallel_traffic" else 2
        rows = n + int(self.robot_visible)
        margin = np.zeros((self.W, rows))
        if model == "orca":
            margin += 0.01 + safety_space
        elif safety_space > 0:
            margin[:, :n] = 0.01 + safety_space
        P = None if model == "orca" else np.tile(sc.default_params(model), (n, 1))
        self._bounds = (proto.traffic_length / 2, proto.traffic_height / 2)
        self.cw = CrowdWorlds(np.zeros((self.W, rows, 13), np.float32), np.full((self.W, n, G, 2), np.nan, np.float32), P, margin,
                              None, type=model, all_params_equal=True, robot_row=self.robot_visible,
                              robot=np.zeros((self.W, 13), np.float32), respawn_bounds=self._bounds if traffic else None,
                              respawn_worlds=np.zeros(self.W, np.int32) if traffic else None)
        seeds = gen.phase_seeds(phase, first_case, self.W, proto.case_capacity)
        self._gen_kw = dict(insert_robot=True, randomize_attributes=proto.randomize_attributes, randomize_positions=True,
                            circle_radius=proto.circle_radius, traffic_length=proto.traffic_length,
                            traffic_height=proto.traffic_height, robot_radius=proto.robot_radius, human_mass=75,
                            robot_mass=robot.mass, robot_desired_speed=robot.desired_speed)
        self._gen_scenario, self._seeds_host = scenario, seeds
        status, scn = gen.generate_worlds(self.cw, scenario, seeds, **self._gen_kw)
        # (the respawn rule stays armed for a hybrid batch even when it drew no traffic world: d_world_flags gates it per
        #  world, and step_device's auto-reset may regenerate any world as a traffic world later)
        if model == "orca":  # RVO2 preferred velocity lives in columns 5:7 (set_state_orca, motion_model_manager.py:105-123)
            S = self.cw.get_states()
            d = S[:, :n, 10:12] - S[:, :n, 0:2]
            dn = np.linalg.norm(d, axis=-1, keepdims=True)
            S[:, :n, 5:7] = np.where(dn > S[:, :n, 12:13], d / np.maximum(dn, 1e-30), d)
            self.cw.set_states(S)
        self.n = n
        self.radius = self.cw.get_states()[:, :n, 8].astype(np.float32)
        self.global_time = np.zeros(self.W, np.float32)
        return self.observe()

    # ------------------------------------------------------------------ device-resident loop (torch tensors in HBM)
    # Every world keeps its next STAGE_DEPTH episodes staged ahead (a power of two; capped so that the staging batch stays below
    # STAGE_BYTES), and every REFILL_EVERY steps one pass of cs_refill_staged_worlds regenerates the consumed slots on a side stream.
    # Measured at 4096 x 25 hybrid worlds with a random policy (tools/gym_step_variants.py; 1 % of the worlds end per step, some of them
    # every five steps -- a robot driven into its neighbour again and again), us per batched step with every finished episode regenerated:
    #   depth 16, pass every 4 steps 95.5 | depth 32, every 16: 65.6 | depth 64, every 32 or 64: 56.6 | depth 128, every 128: 56.1
    #   (no resets at all: 45.5; round 3: 144 same-step, 88 NEXT_STEP).  A pass disturbs the step stream for about its own duration
    #   (~0.2 ms: the generator's wavefronts share SIMDs and the instruction cache with the step kernel), so passes are made rare and
    #   the ring deep; 4096 x 25 worlds take 5.4 MB per level.
    REFILL_EVERY = 32
    STAGE_DEPTH = 64
    STAGE_BYTES = 1 << 30
    REFILL_PRIORITY = 0      # stream priority of the refill passes (1 = the device's lowest: no measurable difference)
    FOLD_RESET = True        # the take-over of the staged episodes inside the step's launch (cs_gym_step_staged) where the step kernel allows it

    def _stage_depth(self):
        rows = self.cw.rows
        level = self.W * (rows * 13 + self.n * self.cw.G * 2 + 13 + 1) * 4
        d = max(1, int(self.STAGE_DEPTH))
        while d > 2 and d * level > self.STAGE_BYTES:
            d //= 2
        return 1 << (d.bit_length() - 1)

    def _device_loop_state(self):
        """Per-world step counters, seeds, the float32 clock table and the pre-staged next episodes, resident on the GPU."""
        import ctypes as C

        import torch

        from .. import generators as gen

        if getattr(self, "_dl", None) is not None and self._dl["cw"] is self.cw:
            return self._dl
        if getattr(self, "_gen_scenario", None) is None:
            raise RuntimeError("step_device needs a world batch generated on the device: reset(..., device=True)")
        if getattr(self, "_dl", None) is not None:       # a new batch: the old one's refill pass may still be writing its staging worlds
            self._release_device_loop()
        try:
            torch.zeros(1, device="cuda")
        except RuntimeError as e:  # _lib.load() maps torch's bundled HIP runtime for both, whatever the import order
            raise _lib.CrowdstepError("torch cannot see the GPU in this process (HIP runtime in use: "
                                      f"{_lib.hip_runtime_path or 'system'}; CROWDSTEP_HIP_RUNTIME=system forces two runtimes): {e}") from e
        W = self.W
        # global_time is accumulated in float32, time_step_factor additions of time_step per step (social_nav_gym.py:244):
        # the same sequence as a table indexed by the per-world step counter
        max_steps = int(self.time_limit / (self.time_step * self.time_step_factor)) + 8
        clock = np.zeros(max_steps, np.float32)
        t = np.float32(0)
        for k in range(1, max_steps):
            for _ in range(self.time_step_factor):
                t = t + np.float32(self.time_step)
            clock[k] = t
        seeds = torch.as_tensor(self._seeds_host.astype(np.int64), device="cuda").to(torch.int32)
        depth = self._stage_depth()
        dl = dict(depth=depth, cw=self.cw, clock=torch.as_tensor(clock, device="cuda"),
                  counter=torch.zeros(W, dtype=torch.int32, device="cuda"),
                  parity=0,
                  results=[(torch.zeros(W, dtype=torch.float32, device="cuda"), torch.zeros(W, dtype=torch.bool, device="cuda"),
                            torch.zeros(W, dtype=torch.bool, device="cuda"), torch.zeros(W, dtype=torch.int32, device="cuda"))
                           for _ in range(2)],
                  gtime=torch.zeros(W, dtype=torch.float32, device="cuda"),
                  seeds=seeds,
                  # a world's seed moves on by the worlds of the WHOLE job when its episode ends: world w walks s + w + k * stride on
                  # any rank count (ShardedBatchedSocialNavGym sets seed_stride = total_worlds; a single process: W)
                  stride=int(getattr(self, "seed_stride", None) or W),
                  mask=torch.zeros(W, dtype=torch.int32, device="cuda"),
                  # pre-staged episodes (include/crowdstep.h cs_stage_book): episode e of world w draws base_seed[w] + e * stride; no slot's
                  # tag is the seed of the episode that belongs in it, so the first refill pass generates every slot
                  base_seed=seeds.clone(), epoch=torch.zeros(W, dtype=torch.int32, device="cuda"),
                  staged_seed=seeds.repeat(depth).contiguous(),
                  staged_status=torch.zeros(W * depth, dtype=torch.int32, device="cuda"),
                  failed=torch.zeros(W, dtype=torch.int32, device="cuda"),
                  pending=torch.zeros(W, dtype=torch.int32, device="cuda"),     # cs_gym_step_staged: take-overs deferred to a later step
                  out=self.cw._buffer("reward_out", (W, 7)).torch(),
                  state=self.cw.d_state.torch().view(W, self.cw.rows, 13),
                  gen=gen.make_generator(self.cw, self._gen_scenario, **self._gen_kw),
                  cols=torch.as_tensor([0, 1, 3, 4, 8] + ([2, 7] if self.headed_obs else []), device="cuda"))
        # the step runs on a torch side stream (ordered against the caller's stream with wait_stream); a library stream carries the
        # refill passes of the staging batch; persistent action / observation tensors
        dl["stream"] = torch.cuda.Stream()
        dl["stream_b"] = _lib.stream_create(priority=self.REFILL_PRIORITY)
        dl["act"] = torch.zeros((W, 2), dtype=torch.float32, device="cuda")
        dl["obs"] = torch.zeros((W, self.n, 7 if self.headed_obs else 5), dtype=torch.float32, device="cuda")
        dl["staging"] = self.cw.staging_copy(depth)
        dl["staging"].stream = dl["stream_b"]
        P = lambda t: t.data_ptr()
        dl["stage_book"] = _lib.cs_stage_book(d_seeds=P(dl["seeds"]), d_base_seed=P(dl["base_seed"]), d_epoch=P(dl["epoch"]),
                                              d_staged_seed=P(dl["staged_seed"]), d_staged_status=P(dl["staged_status"]), d_failed=P(dl["failed"]),
                                              seed_stride=dl["stride"], depth=depth, d_pending=P(dl["pending"]))
        dl["staging_desc"] = dl["staging"].descriptor()
        dl["refill_args"] = (C.byref(dl["gen"]), C.byref(dl["staging_desc"]), C.byref(dl["stage_book"]), C.c_void_p(dl["stream_b"]))
        dl["refill_ev"] = _lib.Event()
        dl["since_refill"] = 0
        self.cw.stream = dl["stream"].cuda_stream
        torch.cuda.synchronize()
        # every world's next STAGE_DEPTH episodes, generated once up front; from here on a pass only regenerates the consumed slots
        _lib.check(_lib.load().cs_refill_staged_worlds(*dl["refill_args"]))
        dl["refill_ev"].record(dl["stream_b"])
        _lib.stream_sync(dl["stream_b"])
        self._dl = dl
        return dl

    def _release_device_loop(self):
        """The refill passes run on a library stream torch's allocator knows nothing about, on tensors torch owns (epoch, base_seed, the
        staging tags): before those tensors can go back to the allocator the stream must have drained -- and it is destroyed, not leaked."""
        dl = getattr(self, "_dl", None)
        if dl is None:
            return
        self._dl = None
        try:
            _lib.stream_sync(dl["stream_b"])
            _lib.stream_destroy(dl["stream_b"])
        except Exception:
            pass

    def close(self):
        """End of the environment's device-resident loop (also run when the object is collected): drains and destroys the refill stream."""
        self._release_device_loop()

    def __del__(self):
        try:
            self._release_device_loop()
        except Exception:
            pass

    def _maybe_refill(self, dl):
        """Every REFILL_EVERY steps: one pass of cs_refill_staged_worlds on the side stream (at most one in flight).  Nothing orders it
        against the step's stream -- the staging slots carry tags (cs_stage_book) -- so it costs the step no wait and no event."""
        dl["since_refill"] += 1
        if dl["since_refill"] < self.REFILL_EVERY or not dl["refill_ev"].done():
            return
        _lib.check(_lib.load().cs_refill_staged_worlds(*dl["refill_args"]))
        dl["refill_ev"].record(dl["stream_b"])
        dl["since_refill"] = 0

    def observe_device(self):
        """Observations [W, N, 5|7] as a torch CUDA tensor gathered from the resident state (no host copy)."""
        dl = self._device_loop_state()
        return dl["state"][:, :self.n].index_select(2, dl["cols"])

    def _gym_book(self, dl, parity, mask, prev_mask, auto_reset):
        reward, terminated, truncated, info = dl["results"][parity]
        return _lib.cs_gym_book(d_counter=dl["counter"].data_ptr(), d_seeds=dl["seeds"].data_ptr(), d_mask=mask.data_ptr(),
                                d_prev_mask=None if prev_mask is None else prev_mask.data_ptr(), d_clock=dl["clock"].data_ptr(),
                                clock_len=dl["clock"].numel(), auto_reset=int(bool(auto_reset)), d_reward=reward.data_ptr(),
                                d_terminated=terminated.data_ptr(), d_truncated=truncated.data_ptr(), d_info=info.data_ptr(),
                                seed_stride=dl["stride"])

    def _step_pieces(self, dl, parity, mode):
        r"""One vectorised Gym step = ONE library launch (cs_gym_step_staged; SFM / HSFM crowds on the one-wavefront kernels) or two on one
        stream, every ctypes argument bound once per (result set, mode):

            cs_gym_step (reward + bookkeeping in the step kernel's prologue, substeps, observation)  ->  cs_consume_staged_worlds

        Which worlds end is known from the reward of the state BEFORE the substeps (social_nav_gym.py:229-233); their next episode
        was generated ahead (its seed is the live one + stride), so the reset is a masked copy.  The regeneration of the consumed
        staging slots (one latency-bound wavefront per world, ~0.15 ms) runs on a side stream with a whole episode to finish
        (_maybe_refill) -- round 3 had it on the critical path (144 us per step) or beside the next step (NEXT_STEP mode, 88 us).

        mode "same_step": the worlds whose episode ends NOW take over their staged episode before the observation is returned;
        mode "next_step" (Gymnasium's default since 1.0): the bookkeeping of this parity writes mask_p and reads prev = mask_{1-p};
        the tail resets the worlds that ended in the PREVIOUS step; mode "none": no tail.

        Plain launches, not a HIP graph: two consecutive replays of a graph leave the stream idle for 8.5 us (rocprofv3 kernel trace,
        profiles/r4ae_gym_step_timeline.txt: 8.6 us between the consume kernel of one replay and the step kernel of the next, 8.4 us
        between one-kernel graphs) where two plain launches follow each other within 0.3 us -- with two launches per step the graph
        saves the host nothing it needs (28 us of host time per step against 44 us on the GPU)."""
        import ctypes as C

        key = ("pieces", parity, mode)
        if key in dl:
            return dl[key]
        cw = self.cw
        A = C.c_void_p(dl["stream"].cuda_stream)
        d = cw.descriptor()
        dref = C.byref(d)
        cfg = (C.c_float * 5)(*[float(x) for x in self.reward_cfg])
        P = lambda t: C.c_void_p(t.data_ptr())
        if mode == "next_step":
            masks = dl["ns_masks"]
            book = self._gym_book(dl, parity, masks[parity], masks[parity ^ 1], True)
            tail_mask = masks[parity ^ 1]
        else:
            book = self._gym_book(dl, parity, dl["mask"], None, mode == "same_step")
            tail_mask = dl["mask"] if mode == "same_step" else None
        # cs_gym_step: reward of the state before the substeps + typed results, step counter, float32 clock, reset mask and next seeds
        # (the head: the step kernel's prologue), the substeps, and the observation of the stepped crowd from the kernel's registers
        a_step = (dref, C.c_float(self.time_step), C.c_int(self.time_step_factor), P(dl["act"]), C.c_float(self.robot_time_step), P(dl["gtime"]), cfg,
                  P(dl["out"]), C.byref(book), C.c_int(int(self.headed_obs)), P(dl["obs"]), A)
        # ... rewritten for the worlds that take over their staged episode (a world whose generation failed keeps its rows and is
        # flagged in reset_failed_mask())
        a_tail = None if tail_mask is None else (C.byref(dl["gen"]), C.byref(dl["staging_desc"]), dref, P(tail_mask), C.byref(dl["stage_book"]),
                                                 C.c_int(int(self.headed_obs)), P(dl["obs"]), A)
        # ... or both in ONE launch (cs_gym_step_staged: the take-over in the step kernel's epilogue) where the step is the one-wavefront
        # SFM / HSFM kernel; FOLD_RESET = False keeps the two launches (the bit-equality reference of the tests)
        a_fold = None
        if tail_mask is not None and self.FOLD_RESET and _lib.load().cs_gym_step_is_one_launch(dref) == 2:
            a_fold = a_step[:-1] + (C.byref(dl["gen"]), C.byref(dl["staging_desc"]), C.byref(dl["stage_book"]), A)
        keep = (d, cfg, book)              # the structs the byref arguments point into
        dl[key] = dict(step=a_step, tail=a_tail, fold=a_fold, keep=keep)
        return dl[key]

    def step_device(self, actions, auto_reset=True):
        """``step`` without leaving the GPU: ``actions`` is a float32 torch CUDA tensor [W, 2] (holonomic vx, vy) -- or
        ``env.action_buffer()`` itself, filled in place (no copy).  Returns torch CUDA tensors (obs [W, N, 5|7], reward [W],
        terminated [W], truncated [W], info_code [W]); obs is a persistent buffer rewritten by the next call, the other four
        alternate between two sets (those of the previous step stay valid for one more call).
        With ``auto_reset=True`` the worlds whose episode ended take over their next episode -- generated ahead on the device from
        the next unused seed -- before the observation is taken (same-step autoreset).  ``auto_reset="next_step"`` is Gymnasium's
        NEXT_STEP mode: a world that ended at step t returns its terminal observation at t and the first observation of its next
        episode -- with reward 0 and no termination -- at t + 1, taken over from the staging batch at the end of step t + 1.
        ``auto_reset=False``: no resets (a finished world keeps stepping)."""
        import torch

        dl = self._device_loop_state()
        mode = "next_step" if auto_reset == "next_step" else ("same_step" if auto_reset else "none")
        if mode == "next_step":
            if "ns_masks" not in dl:
                dl["ns_masks"] = [torch.zeros(self.W, dtype=torch.int32, device="cuda") for _ in range(2)]
                torch.cuda.synchronize()
        elif dl.get("mode") == "next_step" and "ns_masks" in dl and any(bool(m.any().item()) for m in dl["ns_masks"]):
            raise RuntimeError("a NEXT_STEP auto-reset is pending for some world: keep auto_reset=\"next_step\" (or reset()) before changing the mode")
        dl["mode"] = mode
        parity = dl["parity"]
        dl["parity"] ^= 1
        c = self._step_pieces(dl, parity, mode)
        lib, chk = _lib.load(), _lib.check
        side, cur = dl["stream"], torch.cuda.current_stream()
        same = cur.cuda_stream == side.cuda_stream   # the caller already works on the library's stream (`with torch.cuda.stream(env.device_stream())`)
        if not same:
            side.wait_stream(cur)                  # device-side ordering with whatever produced the actions
        if actions is not dl["act"]:
            with torch.cuda.stream(side):
                dl["act"].copy_(actions.to(device="cuda", dtype=torch.float32), non_blocking=True)
        if c["fold"] is not None:
            chk(lib.cs_gym_step_staged(*c["fold"]))
            self._maybe_refill(dl)
        else:
            chk(lib.cs_gym_step(*c["step"]))
            if c["tail"] is not None:
                chk(lib.cs_consume_staged_worlds(*c["tail"]))
                self._maybe_refill(dl)
        if not same:
            cur.wait_stream(side)                  # ... and with whoever reads the results
        reward, terminated, truncated, info = dl["results"][parity]
        return dl["obs"], reward, terminated, truncated, info

    def reset_failed_mask(self):
        """int32 CUDA tensor [W], no synchronisation: 1 where the world's LAST device-side auto-reset could not be generated
        (cs_generate_worlds status != 0: the bounded rejection sampling gave up, or the traffic is too dense -- the reference loops
        forever / raises there).  Such a world keeps its finished rows and starts a new episode FROM them (its step counter and clock
        were reset with the others'): after a collision or a reached goal it ends again at once and tries the following seed of its
        sequence; after a time-limit truncation it runs a further episode from where it stands.  The flag returns to 0 with the first
        reset that succeeds.  2: the take-over was DEFERRED (one-launch step, cs_gym_step_staged: the staged episode was not ready; the
        world is between two episodes -- reward 0, no flags -- until a later step finds it; with the refill cadence of this class that does
        not happen)."""
        return self._device_loop_state()["failed"]

    def failed_resets(self) -> int:
        """Number of worlds flagged in ``reset_failed_mask()``.  Synchronises."""
        return int((self._device_loop_state()["failed"] != 0).sum().item())

    def device_stream(self):
        """The torch stream the device-resident step runs on.  A loop that does its own GPU work inside ``with torch.cuda.stream(env.device_stream())``
        is ordered with the step by the stream itself: step_device then skips its two cross-stream waits (four HIP calls per step)."""
        return self._device_loop_state()["stream"]

    def action_buffer(self):
        """The persistent [W, 2] action tensor the step kernel reads: write actions into it and pass it to ``step_device``."""
        return self._device_loop_state()["act"]

    def lookahead_device(self, action_space):
        """The per-decision array work of CADRL / SARL for every world, on the device: one-step look-ahead of the humans
        (``get_next_human_observable_states``, cadrl.py:258-259 -> cs_peek) and ``compute_rotated_states_and_reward``
        (cadrl.py:42-83 -> cs_lookahead) for all actions.  ``action_space`` [A, 2] (numpy or CUDA tensor).  Returns torch CUDA
        tensors (rotated_states [W, A, N, 13] -- 15 columns with ``headed_obs`` (theta and omega visible) --, rewards [W, A]) ready for ONE
        batched value-network call -- no host copy.
        Needs worlds generated on the device (``reset(..., device=True)``), like ``step_device``."""
        import ctypes as C

        import torch

        dl = self._device_loop_state()
        side, cur_stream = dl["stream"], torch.cuda.current_stream()
        side.wait_stream(cur_stream)
        with torch.cuda.stream(side):               # the library launches on this stream (cw.stream); the torch ops between them follow
            out = self._lookahead_on_side_stream(dl, action_space)
        cur_stream.wait_stream(side)
        return out

    def _lookahead_on_side_stream(self, dl, action_space):
        import ctypes as C

        import torch

        cw, W, n = self.cw, self.W, self.n
        if cw.d_robot is None:
            raise ValueError("lookahead_device needs the robot rows")
        acts = torch.as_tensor(np.asarray(action_space, dtype=np.float32) if not torch.is_tensor(action_space) else action_space,
                               dtype=torch.float32, device="cuda").contiguous()
        A = acts.shape[0]
        lib = _lib.load()
        d = cw.descriptor(respawn=False)
        peek = cw._buffer("peek", (W, n, 8))
        _lib.check(lib.cs_peek(C.byref(d), C.c_float(self.robot_time_step), C.c_void_p(peek.ptr), C.c_void_p(cw.stream)))
        if "la_cols" not in dl:
            # next humans: (px, py, vx, vy), or (x, y, yaw, Vx, Vy, Omega) with theta / omega visible (cadrl.py:42-83) = cs_peek's first six
            dl["la_cols"] = torch.as_tensor([0, 1, 2, 3, 4, 5] if self.headed_obs else [0, 1, 3, 4], device="cuda")
            dl["la_robot_cols"] = torch.as_tensor([0, 1, 3, 4, 8, 10, 11, 12, 2], device="cuda")   # FullState order
            dl["la_robot"] = cw.d_robot.torch().view(W, 13)
        nxt = peek.torch().view(W, n, 8).index_select(2, dl["la_cols"]).contiguous()
        cur = dl["state"][:, :n].index_select(2, dl["cols"]).contiguous()
        rob = dl["la_robot"].index_select(1, dl["la_robot_cols"]).contiguous()
        rot = torch.empty((W, A, n, 15 if self.headed_obs else 13), dtype=torch.float32, device="cuda")
        rew = torch.empty((W, A), dtype=torch.float32, device="cuda")
        _lib.check(lib.cs_lookahead(C.c_int(W), C.c_int(n), C.c_int(A), C.c_int(int(self.headed_obs)), C.c_void_p(acts.data_ptr()),
                                    C.c_void_p(nxt.data_ptr()), C.c_void_p(cur.data_ptr()), C.c_void_p(rob.data_ptr()), C.c_int(9),
                                    C.c_float(self.robot_time_step), C.c_void_p(rot.data_ptr()), C.c_void_p(rew.data_ptr()),
                                    C.c_void_p(cw.stream)))
        return rot, rew

    # ------------------------------------------------------------------ imitation learning, W worlds at once
    def set_human_motion_model_as_robot_policy(self, policy_name, runge_kutta=False, safety_space=0.0):
        """Every world's robot follows a human motion model (SocialNavGym.set_human_motion_model_as_robot_policy,
        social_nav_sim.py:862-873); call after ``reset``.  ``safety_space``: the value the Gym hands to set_safety_space
        (motion_model_manager.py:147-170) -- pass the one used at ``reset``."""
        from .. import scenarios as sc
        from ..batched import HUMAN_MODELS

        if runge_kutta and policy_name == "orca":
            raise NotImplementedError                              # ORCA is Euler only (motion_model_manager.py:579)
        if policy_name not in HUMAN_MODELS:
            raise Exception(f"The robot motion model '{policy_name}' does not exist")
        if self.cw is None:
            raise RuntimeError("reset() first: the robot model is attached to the resident worlds")
        cw, n = self.cw, self.n
        hm = np.zeros((self.W, cw.rows), np.float32)
        if policy_name == "orca":
            hm[:] = 0.01 + safety_space          # robot_sim radii: radius + 0.01 (+ safety space) for humans and robot (:585-587)
            cw.set_robot_model("orca", None, 0.01 + safety_space, hm)
        else:
            # human.safety_space as the single-agent force functions read it: only set_safety_space touches it (:151-153)
            if safety_space > 0 and self._proto.human_policy != "orca":
                hm[:, :n] = 0.01 + safety_space
            cw.set_robot_model(policy_name, sc.default_params(policy_name), (0.01 + safety_space) if safety_space > 0 else 0.0, hm)
        self.robot_motion_model_title = policy_name
        self.robot_runge_kutta = bool(runge_kutta)    # every substep's update_robot is an RK45 solve over dt (cs_robot_model_rk45)

    def imitation_learning_step(self):
        """SocialNavGym.imitation_learning_step (social_nav_gym.py:252-274) for every world: time_step_factor x
        { update_robot ; update_humans }, then reward / termination from the ACTUAL distances of the new state.
        Returns (obs, reward [W], terminated [W], truncated [W], info_code [W])."""
        if getattr(self.cw, "robot_model", None) is None:
            raise AttributeError("set_human_motion_model_as_robot_policy has not been called")
        if getattr(self, "robot_runge_kutta", False):
            # motion_model_manager.py:631-640: the robot's SFM / HSFM model integrated by RK45 over every substep, the humans standing
            # during the solve; then the crowd's Euler substep -- the reference's alternation, launch by launch
            for _ in range(self.time_step_factor):
                self.cw.robot_model_rk45(self.time_step, download=False)
                self.cw.step(self.time_step, 1, None)
        else:
            self.cw.imitation_block(self.time_step, self.time_step_factor)
        for _ in range(self.time_step_factor):
            self.global_time += np.float32(self.time_step)
        out = self.cw.actual_collision_reward(self.robot_time_step, self.global_time, self.reward_cfg)
        return self.observe(), out[:, 3].copy(), out[:, 4] > 0, out[:, 5] > 0, out[:, 6].astype(np.int32)

    def observe(self):
        S = self.cw.get_states()[:, :self.n]
        cols = [0, 1, 3, 4, 8] + ([2, 7] if self.headed_obs else [])
        return S[:, :, cols]

    def step(self, actions):
        """actions [W, 2] holonomic (vx, vy).  Returns (obs, reward [W], terminated [W], truncated [W], info_code [W])."""
        actions = np.ascontiguousarray(np.broadcast_to(np.asarray(actions, np.float32), (self.W, 2)))
        out = self.cw.collision_reward(actions, self.robot_time_step, self.global_time, self.reward_cfg)
        self.cw.step(self.time_step, self.time_step_factor, actions)
        for _ in range(self.time_step_factor):
            self.global_time += np.float32(self.time_step)
        return self.observe(), out[:, 3].copy(), out[:, 4] > 0, out[:, 5] > 0, out[:, 6].astype(np.int32)
