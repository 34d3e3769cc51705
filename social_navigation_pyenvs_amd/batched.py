"""Device-resident batch of W independent crowd worlds driven through the C ABI.

This is the array-level host API above ``libcrowdstep.so``: the batched counterpart of the arrays a
reference ``MotionModelManager`` owns (``states [N,13]``, ``goals [N,G,2]``, ``params [N,20]``,
``obstacles [O,S,2,2]``, ``safety_space [N]``; motion_model_manager.py:253-276) with a leading world
axis W.  Buffers are either owned here (hipMalloc through the ABI) or borrowed from the caller
(``torch.Tensor`` / raw device pointers), so a learner can keep observations on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DeviceBuffer, check, cs_worlds

SFMS = ["sfm_helbing", "sfm_guo", "sfm_moussaid", "hsfm_farina", "hsfm_guo", "hsfm_moussaid",
        "hsfm_new", "hsfm_new_guo", "hsfm_new_moussaid"]  # motion_model_manager.py:15-17
HUMAN_MODELS = SFMS + ["orca"]  # social_nav_gym.py:11-12
# model titles MotionModelManager accepts beyond the Gym's list (motion_model_manager.py:247-251); index = C-ABI type id
CROWD_MODELS = HUMAN_MODELS + ["social_momentum"]
# motion_model_manager.py:14  neighbor_dist, max_neighbors, time_horizon, time_horizon_obstacles
ORCA_DEFAULTS = dict(neighbor_dist=10.0, max_neighbors=10, time_horizon=5.0, time_horizon_obst=5.0)


def _ptr(x) -> int | None:
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):  # torch.Tensor on the GPU
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError("device tensors handed to crowdstep must be contiguous CUDA/HIP tensors")
        return int(x.data_ptr())
    raise TypeError(f"cannot take a device pointer from {type(x)}")


class CrowdWorlds:
    """W worlds x n humans (+ optional robot row) resident in HBM.

    Parameters mirror ``update_humans_parallel`` (forces_parallel.py:185): ``type`` 0..8,
    ``all_params_equal``, ``last_is_robot`` (here ``robot_row``)."""

    def __init__(self, states, goals, params, safety=None, obstacles=None, *, type, all_params_equal=False,
                 robot_row=False, robot=None, respawn_bounds=None, respawn_worlds=None, layout="aos", device=None,
                 stream=None, orca_vertices=None, orca_agent_params=None, orca_math="default"):
        _lib.require_gpu()
        if device is not None:
            _lib.set_device(device)
        if isinstance(type, str):
            type = CROWD_MODELS.index(type)
        if type < 0 or type > 10:
            raise ValueError(f"Type {type} does not exist for this implementation")
        self.type = int(type)
        self.orca = self.type == _lib.CS_ORCA
        self.social_momentum = self.type == _lib.CS_SOCIAL_MOMENTUM
        self.sm_n_actions = 20  # motion_model_manager.py:249
        self.orca_params = dict(ORCA_DEFAULTS)
        self.orca_math = orca_math   # cs_worlds.orca_math of THESE worlds: "default" (CROWDSTEP_ORCA_MATH, else exact) | "exact" | "fast" | "fma"
        self.stream = stream
        states = np.asarray(states, dtype=np.float32)
        if states.ndim == 2:
            states = states[None]
        self.W, self.rows = states.shape[0], states.shape[1]
        self.robot_row = bool(robot_row)
        self.n = self.rows - int(self.robot_row)
        goals = np.asarray(goals, dtype=np.float32)
        if goals.ndim == 3:
            goals = goals[None]
        if goals.shape[0] != self.W or goals.shape[1] != self.n:
            raise ValueError(f"goals shape {goals.shape} does not match W={self.W}, n={self.n}")
        self.G = goals.shape[2]
        if params is None:
            if not (self.orca or self.social_momentum):
                raise ValueError("params are required for the SFM / HSFM models")
            params = np.zeros((self.n, 20), dtype=np.float32)
        params = np.asarray(params, dtype=np.float32)
        self.params_shared = params.ndim == 2
        if params.shape[-2:] != (self.n, 20):
            raise ValueError(f"params must be [..., {self.n}, 20], got {params.shape}")
        if safety is None:
            safety = np.zeros((self.W, self.rows), dtype=np.float32)
        safety = np.ascontiguousarray(np.broadcast_to(np.asarray(safety, dtype=np.float32), (self.W, self.rows)))
        self.layout = layout
        self.all_params_equal = bool(all_params_equal)
        self.respawn_bounds = respawn_bounds
        # ---- device buffers -------------------------------------------------------------
        if layout == "soa":
            st = np.ascontiguousarray(states.reshape(self.W * self.rows, 13).T)
        else:
            st = states
        self.d_state = DeviceBuffer.from_numpy(st)
        self.d_goals = DeviceBuffer.from_numpy(goals)
        self.d_params = DeviceBuffer.from_numpy(params)
        self.d_safety = DeviceBuffer.from_numpy(safety)
        self.O = self.Smax = 0
        self.d_obstacles = None
        self.obstacles_shared = True
        if obstacles is not None:
            obstacles = np.asarray(obstacles, dtype=np.float32)
            self.obstacles_shared = obstacles.ndim == 4
            self.O, self.Smax = obstacles.shape[-4], obstacles.shape[-3]
            self.d_obstacles = DeviceBuffer.from_numpy(obstacles)
        self.d_robot = None
        if robot is not None:
            robot = np.ascontiguousarray(np.broadcast_to(np.asarray(robot, dtype=np.float32), (self.W, 13)))
            self.d_robot = DeviceBuffer.from_numpy(robot)
        self.d_world_flags = None
        if respawn_worlds is not None:  # per-world respawn switch (hybrid batches)
            wf = np.ascontiguousarray(np.broadcast_to(np.asarray(respawn_worlds), (self.W,)).astype(np.int32) & 1)
            self.d_world_flags = DeviceBuffer.from_numpy(wf, dtype=np.int32)
        self.unicycle = False
        self._scratch = {}  # persistent device scratch (no hipMalloc in the stepping loop)
        # ORCA static obstacles: RVO2 vertex records [Nv][8] (rvo2.process_obstacles), shared by all worlds
        self.d_orca_vertices, self.orca_n_vertices = None, 0
        if orca_vertices is not None and len(orca_vertices):
            if not self.orca:
                raise ValueError("orca_vertices only apply to ORCA worlds (SFM / HSFM worlds take `obstacles` segments)")
            ov = np.ascontiguousarray(orca_vertices, dtype=np.float32).reshape(-1, 8)
            self.d_orca_vertices, self.orca_n_vertices = DeviceBuffer.from_numpy(ov), len(ov)
        self.d_orca_agent_params = None
        if orca_agent_params is not None:
            self.set_orca_agent_params(orca_agent_params)

    def set_orca_agent_params(self, agent_params) -> None:
        """RVO2's per-agent neighborDist, maxNeighbors, timeHorizon, timeHorizonObst (RVOSimulator::addAgent's arguments; the reference
        passes ORCA_DEFAULTS for everyone, motion_model_manager.py:241): [W, rows, 4] / [rows, 4], or None = ``orca_params`` for every
        agent.  ``orca_params['max_neighbors']`` / ``['neighbor_dist']`` are raised to the largest per-agent values (the kernel's
        neighbour columns and the grid's cell edge are laid out for them)."""
        if agent_params is None:
            self.d_orca_agent_params = None
            return
        if not self.orca:
            raise ValueError("per-agent RVO2 parameters only apply to ORCA worlds")
        ap = np.ascontiguousarray(np.broadcast_to(np.asarray(agent_params, dtype=np.float32), (self.W, self.rows, 4)))
        if np.any(ap[..., 1] < 0) or np.any(ap[..., 1] > 16) or np.any(ap[..., 2] <= 0) or np.any(ap[..., 0] < 0):
            raise ValueError("per-agent RVO2 parameters: 0 <= maxNeighbors <= 16, timeHorizon > 0, neighborDist >= 0")
        self.orca_params = dict(self.orca_params, max_neighbors=int(max(self.orca_params["max_neighbors"], ap[..., 1].max())),
                                neighbor_dist=float(max(self.orca_params["neighbor_dist"], ap[..., 0].max())))
        self.d_orca_agent_params = DeviceBuffer.from_numpy(ap)

    # ------------------------------------------------------------------ descriptor
    def _flags(self, respawn=None) -> int:
        f = 0
        if self.all_params_equal:
            f |= _lib.CS_ALL_PARAMS_EQUAL
        if self.robot_row:
            f |= _lib.CS_ROBOT_ROW
        if self.params_shared:
            f |= _lib.CS_PARAMS_SHARED
        if self.obstacles_shared:
            f |= _lib.CS_OBSTACLES_SHARED
        if (self.respawn_bounds is not None) if respawn is None else respawn:
            f |= _lib.CS_RESPAWN
        if self.unicycle:
            f |= _lib.CS_ROBOT_UNICYCLE
        return f

    def descriptor(self, respawn=None) -> cs_worlds:
        d = cs_worlds()
        d.W, d.n, d.G, d.O, d.Smax = self.W, self.n, self.G, self.O, self.Smax
        d.type = self.type
        d.flags = self._flags(respawn)
        d.layout = _lib.CS_LAYOUT_SOA if self.layout == "soa" else _lib.CS_LAYOUT_AOS
        d.d_state = _ptr(self.d_state)
        d.d_goals = _ptr(self.d_goals)
        d.d_params = _ptr(self.d_params)
        d.d_safety = _ptr(self.d_safety)
        d.d_obstacles = _ptr(self.d_obstacles)
        d.d_robot = _ptr(self.d_robot)
        d.d_world_flags = _ptr(self.d_world_flags)
        bx, by = self.respawn_bounds if self.respawn_bounds is not None else (0.0, 0.0)
        d.respawn_bound_x, d.respawn_bound_y = float(bx), float(by)
        d.orca_neighbor_dist = float(self.orca_params["neighbor_dist"])
        d.orca_max_neighbors = int(self.orca_params["max_neighbors"])
        d.orca_time_horizon = float(self.orca_params["time_horizon"])
        d.orca_time_horizon_obst = float(self.orca_params["time_horizon_obst"])
        d.sm_n_actions = int(self.sm_n_actions)
        d.d_orca_vertices = _ptr(self.d_orca_vertices)
        d.orca_n_vertices = int(self.orca_n_vertices)
        d.d_orca_agent_params = _ptr(getattr(self, "d_orca_agent_params", None))
        if self.orca_math not in _lib.ORCA_MATH_NAMES:
            raise ValueError(f"orca_math must be one of {sorted(_lib.ORCA_MATH_NAMES)}, got {self.orca_math!r}")
        d.orca_math = _lib.ORCA_MATH_NAMES[self.orca_math]
        return d

    # ------------------------------------------------------------------ hot path
    def update_humans_parallel(self, dt: float, in_place: bool = True):
        """One substep, semantics of forces_parallel.py:185.  Returns the device buffer holding the
        updated rows (the state buffer itself when ``in_place``)."""
        d = self.descriptor(respawn=False)
        out = self.d_state if in_place else DeviceBuffer(self.d_state.shape)
        check(_lib.load().cs_update_humans_parallel(C.byref(d), C.c_float(dt), C.c_void_p(_ptr(out)),
                                                    C.c_void_p(self.stream)))
        return out

    def step(self, dt: float, n_substeps: int = 1, action=None) -> None:
        """n_substeps fused substeps (robot move + update_humans + respawn), in place."""
        d = self.descriptor()
        a_ptr = None
        if action is not None:
            if isinstance(action, np.ndarray) or isinstance(action, (list, tuple)):
                a_ptr = self._upload("action", np.broadcast_to(np.asarray(action, dtype=np.float32), (self.W, 2))).ptr
            else:
                a_ptr = _ptr(action)
        check(_lib.load().cs_step(C.byref(d), C.c_float(dt), C.c_int(n_substeps), C.c_void_p(a_ptr),
                                  C.c_void_p(self.stream)))

    def step_trace(self, dt: float, n_substeps: int = 1, action=None) -> np.ndarray:
        """step() through cs_step_trace: the same kernel build and arithmetic, and every human's row after every fused substep:
        [n_substeps, W, rows, 12] = px, py, theta, vx, vy, bvx, bvy, omega, gx, gy, goals[0].x, goals[0].y (the robot row, if any,
        as the following substep sees it)."""
        d = self.descriptor()
        a_ptr = None
        if isinstance(action, (np.ndarray, list, tuple)):
            a_ptr = self._upload("action", np.broadcast_to(np.asarray(action, dtype=np.float32), (self.W, 2))).ptr
        a_ptr = a_ptr if (action is None or isinstance(action, (np.ndarray, list, tuple))) else _ptr(action)
        out = self._buffer("trace", (int(n_substeps), self.W, self.rows, 12))
        check(_lib.load().cs_step_trace(C.byref(d), C.c_float(dt), C.c_int(n_substeps), C.c_void_p(a_ptr), C.c_void_p(out.ptr),
                                        C.c_void_p(self.stream)))
        return out.download(self.stream)

    def _buffer(self, name, shape, dtype=np.float32) -> DeviceBuffer:
        buf = self._scratch.get(name)
        if buf is None or buf.shape != tuple(shape) or buf.dtype != np.dtype(dtype):
            buf = DeviceBuffer(tuple(shape), dtype)
            self._scratch[name] = buf
        return buf

    def _upload(self, name, arr, dtype=np.float32) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr, dtype=dtype)
        buf = self._buffer(name, arr.shape, dtype)
        buf.upload(arr, self.stream)
        return buf

    def peek(self, dt: float) -> np.ndarray:
        """[W, n, 8] rows x, y, yaw, Vx, Vy, Omega, Gx, Gy of the next state; nothing is committed."""
        d = self.descriptor(respawn=False)
        out = self._buffer("peek", (self.W, self.n, 8))
        check(_lib.load().cs_peek(C.byref(d), C.c_float(dt), C.c_void_p(out.ptr), C.c_void_p(self.stream)))
        return out.download(self.stream)

    def collision_reward(self, action, T: float, global_time, reward_cfg=(50.0, 1.0, -0.25, 0.2, 0.5)) -> np.ndarray:
        """[W, 7] = collision, dmin, reaching_goal, reward, terminated, truncated, info_code."""
        if self.d_robot is None:
            raise ValueError("collision_reward needs the robot rows")
        d = self.descriptor()
        act = self._upload("reward_action", np.broadcast_to(np.asarray(action, dtype=np.float32), (self.W, 2)))
        gt = self._upload("global_time", np.broadcast_to(np.asarray(global_time, dtype=np.float32), (self.W,)))
        out = self._buffer("reward_out", (self.W, 7))
        cfg = (C.c_float * 5)(*[float(x) for x in reward_cfg])
        check(_lib.load().cs_collision_reward(C.byref(d), C.c_void_p(act.ptr), C.c_float(T), C.c_void_p(gt.ptr),
                                              cfg, C.c_void_p(out.ptr), C.c_void_p(self.stream)))
        return out.download(self.stream)

    def laser_scan(self, range_: float, samples: int, max_distance: float, pose=None) -> np.ndarray:
        """[W, samples] laser distances of every world's sensor (LaserSensor.get_laser_measurements without noise,
        sensors.py:51-66).  ``pose`` [W, 3] (x, y, yaw) or None = the robot rows."""
        if max_distance > 10:
            raise ValueError("Maxium distance for laser is 10 meters")
        d = self.descriptor()
        p_ptr = None
        if pose is not None:
            p_ptr = self._upload("laser_pose", np.broadcast_to(np.asarray(pose, dtype=np.float32), (self.W, 3))).ptr
        out = self._buffer("laser_out", (self.W, int(samples)))
        check(_lib.load().cs_laser_scan(C.byref(d), C.c_void_p(p_ptr), C.c_int(3), C.c_float(range_), C.c_int(int(samples)),
                                        C.c_float(max_distance), C.c_void_p(out.ptr), C.c_void_p(self.stream)))
        return out.download(self.stream)

    # ------------------------------------------------------------------ RK45 integration (runge_kutta=True)
    def update_humans_rk45(self, dt: float, desired_force=None) -> np.ndarray:
        """MotionModelManager(runge_kutta=True).update_humans(t, dt) (motion_model_manager.py:374-384) of every world: one
        adaptive RK45 solve over dt, in place.  ``desired_force`` [W, n, 2]: the humans' desired_force attributes to start
        from (None = what the previous call left, zeros at first).  Returns the number of right-hand-side evaluations [W]."""
        if self.type > 8:
            raise ValueError(f"Type {self.type} does not exist for this implementation")
        mem = getattr(self, "d_rk_memory", None)
        if mem is None:
            mem = self.d_rk_memory = DeviceBuffer.from_numpy(np.zeros((self.W, self.n, 2), np.float32))
        if desired_force is not None:
            mem.upload(np.ascontiguousarray(np.broadcast_to(np.asarray(desired_force, dtype=np.float32), (self.W, self.n, 2))), self.stream)
        d = self.descriptor()     # (with respawn bounds set: the parallel-traffic respawn rule runs behind the solve)
        nfev = self._buffer("rk_nfev", (self.W,), np.int32)
        check(_lib.load().cs_update_humans_rk45(C.byref(d), C.c_float(dt), C.c_void_p(mem.ptr), C.c_void_p(nfev.ptr),
                                                C.c_void_p(self.stream)))
        return nfev.download(self.stream)

    def complete_rk45_simulation(self, dt: float, final_time: float, n_eval: int, desired_force=None):
        """MotionModelManager.complete_rk45_simulation(t, dt, final_time) (motion_model_manager.py:461-498) of every world: ONE
        adaptive RK45 solve over final_time, the solution at the ``n_eval`` times k * dt from the solver's dense output.
        Returns (human_states [W, n_eval, n, 6 | 4], nfev [W]); the rows are left at the final state."""
        if self.type > 8:
            raise ValueError(f"Type {self.type} does not exist for this implementation")
        mem = getattr(self, "d_rk_memory", None)
        if mem is None:
            mem = self.d_rk_memory = DeviceBuffer.from_numpy(np.zeros((self.W, self.n, 2), np.float32))
        if desired_force is not None:
            mem.upload(np.ascontiguousarray(np.broadcast_to(np.asarray(desired_force, dtype=np.float32), (self.W, self.n, 2))), self.stream)
        d = self.descriptor(respawn=False)
        ns = 6 if self.type >= 3 else 4
        out = self._buffer("rk_dense", (self.W, int(n_eval), self.n, ns))
        nfev = self._buffer("rk_nfev", (self.W,), np.int32)
        check(_lib.load().cs_complete_rk45_simulation(C.byref(d), C.c_float(dt), C.c_float(final_time), C.c_void_p(mem.ptr), C.c_void_p(out.ptr),
                                                      C.c_int(int(n_eval)), C.c_void_p(nfev.ptr), C.c_void_p(self.stream)))
        return out.download(self.stream), nfev.download(self.stream)

    def robot_model_rk45(self, dt: float, download: bool = True):
        """update_robot(t, dt) of a robot whose SFM / HSFM model is integrated with RK45 (motion_model_manager.py:631-640), every world,
        in place on the robot rows.  Returns the number of right-hand-side evaluations [W] (``download=False``: None, no
        synchronisation -- the loop of BatchedSocialNavGym.imitation_learning_step)."""
        if getattr(self, "robot_model", None) is None:
            raise ValueError("no robot motion model set")
        d = self.descriptor()
        pr = (C.c_float * 20)(*[float(x) for x in self.robot_params])
        nfev = self._buffer("robot_rk_nfev", (self.W,), np.int32)
        check(_lib.load().cs_robot_model_rk45(C.byref(d), C.c_int(self.robot_model), pr, C.c_float(self.robot_margin),
                                              C.c_void_p(_ptr(self.d_human_margin)), C.c_void_p(_ptr(self.d_robot_memory)), C.c_float(dt),
                                              C.c_void_p(nfev.ptr), C.c_void_p(self.stream)))
        return nfev.download(self.stream) if download else None

    # ------------------------------------------------------------------ the robot under a human motion model
    def set_robot_model(self, model, params=None, margin=0.0, human_margin=None, orca_vertices=None) -> None:
        """MotionModelManager.set_robot_motion_model (motion_model_manager.py:552-589) for every world: the robot rows are
        stepped by ``model`` (one of HUMAN_MODELS: nine SFM / HSFM titles or "orca") instead of an action.
        ``params``: the robot's 20 parameters (SFM / HSFM); ``margin``: robot.safety_space as its model adds it to the radius;
        ``human_margin`` [W, rows] / scalar: the humans' safety space as the robot's model sees it (None = the crowd's own);
        ``orca_vertices``: RVO2 vertex records of the walls for an ORCA robot (rvo2.process_obstacles)."""
        if self.d_robot is None:
            raise ValueError("set_robot_model needs the robot rows")
        if isinstance(model, str):
            if model not in HUMAN_MODELS:
                raise Exception(f"The robot motion model '{model}' does not exist")  # motion_model_manager.py:589
            model = HUMAN_MODELS.index(model)
        self.robot_model = int(model)
        self.robot_params = np.zeros(20, np.float32) if params is None else np.ascontiguousarray(params, dtype=np.float32).reshape(20)
        self.robot_margin = float(margin)
        self.d_human_margin = None
        if human_margin is not None:
            hm = np.ascontiguousarray(np.broadcast_to(np.asarray(human_margin, dtype=np.float32), (self.W, self.rows)))
            self.d_human_margin = DeviceBuffer.from_numpy(hm)
        self.d_robot_memory = DeviceBuffer.from_numpy(np.zeros((self.W, 2), np.float32))
        if orca_vertices is not None and len(orca_vertices):
            ov = np.ascontiguousarray(orca_vertices, dtype=np.float32).reshape(-1, 8)
            self.d_orca_vertices, self.orca_n_vertices = DeviceBuffer.from_numpy(ov), len(ov)

    def robot_model_step(self, dt: float, just_velocities: bool = False) -> None:
        """update_robot(t, dt, just_velocities) of every world (motion_model_manager.py:615-653), in place on the robot rows."""
        if getattr(self, "robot_model", None) is None:
            raise ValueError("no robot motion model set")
        d = self.descriptor()
        pr = (C.c_float * 20)(*[float(x) for x in self.robot_params])
        fn = _lib.load().cs_robot_model_velocities if just_velocities else _lib.load().cs_robot_model_step
        check(fn(C.byref(d), C.c_int(self.robot_model), pr, C.c_float(self.robot_margin),
                                              C.c_void_p(_ptr(self.d_human_margin)), C.c_void_p(_ptr(self.d_robot_memory)),
                                              C.c_float(dt), C.c_void_p(self.stream)))

    def imitation_block(self, dt: float, n_substeps: int, graph: bool = True) -> None:
        """The substep loop of SocialNavGym.imitation_learning_step (social_nav_gym.py:259-263):
        n_substeps x { update_robot(t, dt) ; update_humans(t, dt) } = cs_imitation_block: two launches when the robot is invisible
        to the crowd (the crowd's fused substeps record what the robot sees, the robot integrates behind them), the reference's
        strict alternation of 2 x n_substeps launches otherwise.  ``graph=False``: always the alternating launches."""
        if getattr(self, "robot_model", None) is None:
            raise ValueError("no robot motion model set")
        if not graph:
            for _ in range(int(n_substeps)):
                self.robot_model_step(dt)
                self.step(dt, 1, None)
            return
        d = self.descriptor()
        pr = (C.c_float * 20)(*[float(x) for x in self.robot_params])
        check(_lib.load().cs_imitation_block(C.byref(d), C.c_int(self.robot_model), pr, C.c_float(self.robot_margin),
                                             C.c_void_p(_ptr(self.d_human_margin)), C.c_void_p(_ptr(self.d_robot_memory)),
                                             C.c_float(dt), C.c_int(int(n_substeps)), C.c_void_p(self.stream)))

    def reserve_scratch(self, n_substeps: int = 20) -> None:
        """cs_reserve_scratch: allocate, outside any stream capture, the library-owned scratch that imitation_block (fused form) or
        step (worlds beyond one block) need for this batch on its stream -- required before the first such call is CAPTURED."""
        d = self.descriptor()
        check(_lib.load().cs_reserve_scratch(C.byref(d), C.c_int(int(n_substeps)), C.c_void_p(self.stream)))

    def actual_collision_reward(self, T: float, global_time, reward_cfg=(50.0, 1.0, -0.25, 0.2, 0.5)) -> np.ndarray:
        """[W, 7] like collision_reward, from the distances of the current state (social_nav_gym.py:107-118)."""
        if self.d_robot is None:
            raise ValueError("actual_collision_reward needs the robot rows")
        d = self.descriptor()
        gt = self._upload("global_time", np.broadcast_to(np.asarray(global_time, dtype=np.float32), (self.W,)))
        out = self._buffer("reward_out", (self.W, 7))
        cfg = (C.c_float * 5)(*[float(x) for x in reward_cfg])
        check(_lib.load().cs_actual_collision_reward(C.byref(d), C.c_float(T), C.c_void_p(gt.ptr), cfg, C.c_void_p(out.ptr),
                                                     C.c_void_p(self.stream)))
        return out.download(self.stream)

    # ------------------------------------------------------------------ state access
    def sync(self):
        _lib.stream_sync(self.stream)

    def get_states(self, buf=None) -> np.ndarray:
        arr = (buf or self.d_state).download(self.stream)
        if self.layout == "soa":
            arr = arr.reshape(13, self.W * self.rows).T
        return np.ascontiguousarray(arr).reshape(self.W, self.rows, 13)

    def set_states(self, states) -> None:
        states = np.asarray(states, dtype=np.float32).reshape(self.W, self.rows, 13)
        if self.layout == "soa":
            states = np.ascontiguousarray(states.reshape(self.W * self.rows, 13).T)
        self.d_state.upload(states, self.stream)

    def get_goals(self) -> np.ndarray:
        return self.d_goals.download(self.stream)

    def set_goals(self, goals) -> None:
        self.d_goals.upload(np.asarray(goals, dtype=np.float32).reshape(self.W, self.n, self.G, 2), self.stream)

    def get_robot(self) -> np.ndarray:
        return self.d_robot.download(self.stream)

    def set_robot(self, robot) -> None:
        robot = np.ascontiguousarray(np.broadcast_to(np.asarray(robot, dtype=np.float32), (self.W, 13)))
        if self.d_robot is None:
            self.d_robot = DeviceBuffer.from_numpy(robot)
        else:
            self.d_robot.upload(robot, self.stream)

    def set_safety(self, safety) -> None:
        safety = np.ascontiguousarray(np.broadcast_to(np.asarray(safety, dtype=np.float32), (self.W, self.rows)))
        self.d_safety.upload(safety, self.stream)

    def launch_geometry(self):
        d = self.descriptor()
        g, b, wpb = C.c_int(), C.c_int(), C.c_int()
        check(_lib.load().cs_launch_geometry(C.byref(d), C.byref(g), C.byref(b), C.byref(wpb)))
        return g.value, b.value, wpb.value

    def step_variant(self, entry: str = "step") -> str:
        """Name of the kernel build the library runs for these worlds (cs_step_variant): "step" = cs_step,
        "update" = cs_update_humans_parallel out of place, "peek" = cs_peek."""
        d = self.descriptor(respawn=False if entry != "step" else None)
        buf = C.create_string_buffer(256)
        check(_lib.load().cs_step_variant(C.byref(d), C.c_int({"step": 0, "update": 1, "peek": 2}[entry]), buf, C.c_size_t(256)))
        return buf.value.decode()

    def staging_copy(self, depth: int = 1):
        """A second batch of the same shape -- ``depth`` times the worlds (world j * W + w = slot j of world w: the staging batch of
        the pre-staged episodes, include/crowdstep.h cs_stage_book) -- whose world-dependent buffers (state, goals, robot rows,
        world flags) are separate allocations and everything else (parameters, margins, walls) is shared: a target of the device
        generators; cs_copy_worlds_masked / cs_consume_staged_worlds move the generated worlds over afterwards.  (With depth > 1 the
        shared per-world buffers do not cover it: such a batch is for the generators and the copies only, never stepped.)"""
        import copy

        st = copy.copy(self)
        st._scratch = {}
        st.W = self.W * int(depth)
        grow = lambda shape: (shape[0] * int(depth),) + tuple(shape[1:])
        if self.layout == "soa":      # planes [13][W * rows]
            st.d_state = DeviceBuffer((self.d_state.nbytes // 4 * int(depth),))
        else:
            st.d_state = DeviceBuffer(grow(self.d_state.shape))
        st.d_goals = DeviceBuffer(grow(self.d_goals.shape))
        st.d_robot = None if self.d_robot is None else DeviceBuffer(grow(self.d_robot.shape))
        st.d_world_flags = None if self.d_world_flags is None else DeviceBuffer(grow(self.d_world_flags.shape), np.int32)
        return st
