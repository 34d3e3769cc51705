"""Agent / HumanAgent / RobotAgent state records (reference: social_gym/src/agent.py, human_agent.py,
robot_agent.py) without any sprite / PyGame part: rendering is bypassed on this path.

Attribute names, ``get_safe_state`` row layout (agent.py:256-258), the 20-slot parameter row
(agent.py:268-388), ``set_state`` (numpy *views* into the manager's state rows, agent.py:260-266) and the
robot's ``set / step / compute_position / act / configure`` follow the reference so CrowdNav code that
mutates these objects keeps working."""
from __future__ import annotations

import logging

import numpy as np

from ...crowd_nav.utils.action import ActionRot, ActionXY
from ...crowd_nav.utils.state import FullState, JointState, ObservableState, ObservableStateHeaded
from .utils import PRECISION

# slot -> attribute name of the parameter row (agent.py:269)
PARAM_SLOTS = ["relaxation_time", "Ai", "Aw", "Bi", "Bw", "Ci", "Cw", "Di", "Dw", "Ei", "k1", "k2", "agent_lambda",
               "gamma", "ns", "ns1", "ko", "kd", "alpha", "k_lambda"]
# model constants (agent.py:94-243)
_DEFAULTS = dict(relaxation_time=0.5, Ai=2000.0, Aw=2000.0, Bi=0.08, Bw=0.08, Ci=120.0, Cw=120.0, Di=0.6, Dw=0.6,
                 Ei=360, k1=120000.0, k2=240000.0, agent_lambda=2.0, gamma=0.35, ns=2.0, ns1=3.0, ko=1.0, kd=500.0,
                 alpha=3.0, k_lambda=0.1)
_SOCIAL = {
    "helbing": ["relaxation_time", "Ai", "Aw", "Bi", "Bw", "k1", "k2"],
    "guo": ["relaxation_time", "Ai", "Aw", "Bi", "Bw", "Ci", "Cw", "Di", "Dw", "k1", "k2"],
    "moussaid": ["relaxation_time", "Ei", "agent_lambda", "gamma", "ns", "ns1", "Aw", "Bw", "k1", "k2"],
}
_HEADED = ["ko", "kd", "alpha", "k_lambda"]


def model_param_names(model: str) -> list:
    """Attributes the reference defines (and packs) for `model`."""
    if model.endswith("guo"):
        names = list(_SOCIAL["guo"])
    elif model.endswith("moussaid"):
        names = list(_SOCIAL["moussaid"])
    elif model in ("sfm_helbing", "hsfm_farina", "hsfm_new"):
        names = list(_SOCIAL["helbing"])
    else:
        return []
    if model.startswith("hsfm"):
        names += _HEADED
    return names


class Agent:
    def __init__(self, position, yaw, color=(0, 0, 0), radius=0.3, real_size=15, display_ratio=1.0, mass=80, desired_speed=1):
        self.position = np.array(position, dtype=PRECISION)
        self.yaw = yaw
        self.color = color
        self.radius = radius
        self.safety_space = 0
        self.obstacles = []
        self.mass = mass
        self.desired_speed = desired_speed
        self.linear_velocity = np.array([0.0, 0.0], dtype=PRECISION)
        self.body_velocity = np.array([0.0, 0.0], dtype=PRECISION)
        self.angular_velocity = 0.0
        self.headed = False
        self.orca = False
        self.inertia = 0.5 * self.mass * self.radius * self.radius
        self.rotational_matrix = np.zeros((2, 2), dtype=PRECISION)
        self.goals = list()
        self.policy = None
        self.kinematics = None
        self.sensor = None
        self.visible = None

    # -- geometry ---------------------------------------------------------------------------------
    def get_goal_position(self):
        return np.array([self.goals[0][0], self.goals[0][1]], dtype=PRECISION)

    def get_position(self):
        return self.position

    def get_pose(self):
        return np.append(self.position, self.yaw)

    def set_pose(self, pose):
        self.position = pose[0:2]
        self.yaw = pose[2]

    def set_goals(self, goals):
        self.goals = goals

    def compute_rotational_matrix(self):
        c, s = np.cos(self.yaw), np.sin(self.yaw)
        self.rotational_matrix = np.array([[c, -s], [s, c]], dtype=PRECISION)

    # rendering hooks of the reference are no-ops here
    def update(self):
        pass

    def move(self):
        pass

    def rotate(self):
        pass

    def render(self, *a, **k):
        pass

    # -- model parameters -------------------------------------------------------------------------
    def set_parameters(self, model: str):
        for name in model_param_names(model):
            setattr(self, name, _DEFAULTS[name])

    def get_parameters(self, model: str):
        params = np.zeros((20,), PRECISION)
        for name in model_param_names(model):
            params[PARAM_SLOTS.index(name)] = getattr(self, name)
        return params

    # -- CrowdNav views ---------------------------------------------------------------------------
    def get_observable_state(self, visible_theta_and_omega=False):
        if visible_theta_and_omega:
            return ObservableStateHeaded(self.position[0], self.position[1], self.linear_velocity[0],
                                         self.linear_velocity[1], self.radius, self.yaw, self.angular_velocity)
        return ObservableState(self.position[0], self.position[1], self.linear_velocity[0], self.linear_velocity[1],
                               self.radius)

    def get_full_state(self):
        return FullState(self.position[0], self.position[1], self.linear_velocity[0], self.linear_velocity[1],
                         self.radius, self.goals[0][0], self.goals[0][1], self.desired_speed, self.yaw)

    # -- array seam -------------------------------------------------------------------------------
    def get_safe_state(self):
        return np.array([*np.copy(self.position), self.yaw, *np.copy(self.linear_velocity),
                         *np.copy(self.body_velocity), self.angular_velocity, self.radius, self.mass,
                         *self.goals[0], self.desired_speed], PRECISION)

    def set_state(self, pose_and_velocity):
        # [px, py, theta, vx, vy, bvx, bvy, omega] -- slices stay views of the caller's row
        self.position = pose_and_velocity[0:2]
        self.yaw = pose_and_velocity[2]
        self.linear_velocity = pose_and_velocity[3:5]
        self.body_velocity = pose_and_velocity[5:7]
        self.angular_velocity = pose_and_velocity[7]


class HumanAgent(Agent):
    def __init__(self, game, label, model, pos, yaw, goals, color=(0, 0, 0), radius=0.3, mass=80, des_speed=1, group_id=-1):
        super().__init__(pos, yaw, color, radius, mass=mass, desired_speed=des_speed)
        self.motion_model = model
        self.group_id = group_id
        self.goals = goals
        self.incremental_index = label
        self.set_parameters(self.motion_model)


class RobotAgent(Agent):
    def __init__(self, game=None, pos=(7.5, 7.5), yaw=0.0, radius=0.3, goals=None, mass=80, desired_speed=1):
        super().__init__(np.array(pos, dtype=PRECISION), yaw, (255, 0, 0), radius, mass=mass, desired_speed=desired_speed)
        self.goals = goals if goals is not None else list()
        self.collisions = 0
        self.laser = None
        self.parallelize = False
        self.time_step = None

    def set_radius_and_update_graphics(self, radius):
        self.radius = radius

    # -- laser range finder (robot_agent.py:67-82; ray casting on the GPU, csrc/laser.hip) ----------
    def add_laser_sensor(self, range: float, samples: int, max_distance: float, uncertainty=None, render=False):
        from .sensors import LaserSensor

        self.laser = LaserSensor(self.position, self.yaw, range, samples, max_distance, uncertainty=uncertainty)
        self.laser_data = {}
        self.laser_render = render

    def get_laser_readings(self, humans: list, walls):
        self.laser.update_pose(self.position, self.yaw)
        readings = self.laser.get_laser_measurements(humans, walls)
        # the laser sits at the robot centre: readings are reported from the robot's surface (:80-81)
        self.laser_data = {k: v - self.radius for k, v in readings.items()}
        return self.laser_data

    # -- CrowdNav interface (robot_agent.py:93-160) -------------------------------------------------
    def set_policy(self, policy):
        self.policy = policy
        self.kinematics = policy.kinematics
        if "hsfm" in policy.name:
            self.headed = True
        if policy.name == "orca":
            self.orca = True

    def set(self, px, py, gx, gy, vx, vy, theta, radius=None, v_pref=None, w=None):
        self.position[0] = px
        self.position[1] = py
        self.goals.insert(0, [gx, gy])
        if len(self.goals) > 1:
            self.goals.pop()
        self.goals[0] = [gx, gy]
        self.linear_velocity[0] = vx
        self.linear_velocity[1] = vy
        self.yaw = theta
        if radius is not None:
            self.radius = radius
        if v_pref is not None:
            self.desired_speed = v_pref
        if w is not None:
            self.angular_velocity = w

    def check_validity(self, action):
        if self.kinematics == "holonomic":
            assert isinstance(action, ActionXY)
        else:
            assert isinstance(action, ActionRot)

    def compute_position(self, action, delta_t):
        self.check_validity(action)
        if self.kinematics == "holonomic":
            act = np.array([action.vx, action.vy], dtype=PRECISION)
        else:
            act = np.array([np.cos(self.yaw + action.r) * action.v, np.sin(self.yaw + action.r) * action.v], dtype=PRECISION)
        return self.position + act * delta_t

    def step(self, action, delta_t):
        self.check_validity(action)
        self.position = self.compute_position(action, delta_t)
        if self.kinematics == "holonomic":
            self.linear_velocity = np.array([action.vx, action.vy], dtype=PRECISION)
        else:
            self.yaw = (self.yaw + action.r) % (2 * np.pi)
            self.linear_velocity = np.array([np.cos(self.yaw) * action.v, np.sin(self.yaw) * action.v], dtype=PRECISION)

    def act(self, ob):
        if self.policy is None:
            raise AttributeError("Policy attribute has to be set!")
        return self.policy.predict(JointState(self.get_full_state(), ob))

    def configure(self, config, section, policy_factory=None):
        """Reads [robot] visible / v_pref / radius / policy / sensor (robot_agent.py:144-150).  The policy
        classes are consumers of this env, not part of it: pass the caller's `policy_factory` mapping."""
        self.visible = config.getboolean(section, "visible")
        self.desired_speed = config.getfloat(section, "v_pref")
        self.radius = config.getfloat(section, "radius")
        name = config.get(section, "policy")
        self.policy = policy_factory[name]() if (policy_factory is not None and name in policy_factory) else None
        self.sensor = config.get(section, "sensor")
        self.kinematics = self.policy.kinematics if self.policy is not None else None

    def print_info(self):
        logging.info("Agent is {} and has {} kinematic constraint".format("visible" if self.visible else "invisible", self.kinematics))
