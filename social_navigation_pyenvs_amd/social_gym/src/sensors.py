"""LaserSensor -- host mirror of the reference class over ``cs_laser_scan``.

Reference: /root/reference/social_gym/src/sensors.py:6-71.  Same constructor, ``update_pose``,
``get_laser_measurements(humans, walls) -> {angle: distance}`` and ``add_uncertainty``; the ray casting itself runs
on the GPU (``csrc/laser.hip``), the optional Gaussian noise is drawn on the host from numpy's global stream in ray
order, as the reference does (:65, :69-71).

Deliberate deviation: the reference leaves ``self.uncertainty`` unset when the constructor gets ``uncertainty=None``
(:15) and ``get_laser_measurements`` then raises ``AttributeError`` (:65); here ``None`` means "no noise".
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ... import _lib
from ..._lib import DeviceBuffer, check, cs_worlds
from .utils import PRECISION


def walls_to_array(walls) -> np.ndarray | None:
    """[O][Smax][2][2] NaN-padded segment array from objects with a ``segments`` dict (obstacle.py:31-32)."""
    if not walls:
        return None
    segs = [list(w.segments.values()) for w in walls]
    smax = max(len(s) for s in segs)
    arr = np.full((len(segs), smax, 2, 2), np.nan, dtype=np.float32)
    for i, ss in enumerate(segs):
        for j, sg in enumerate(ss):
            arr[i, j] = np.asarray(sg, dtype=np.float32)
    return arr


def laser_scan(states, pose, range_, samples, max_distance, obstacles=None, stream=None) -> np.ndarray:
    """Batched form: states [W][n][13] (x, y, .., radius in column 8), pose [W][3] -> distances [W][samples]."""
    _lib.require_gpu()
    if max_distance > 10:
        raise ValueError("Maxium distance for laser is 10 meters")
    states = np.ascontiguousarray(states, dtype=np.float32)
    W, n = states.shape[0], states.shape[1]
    d = cs_worlds()
    d.W, d.n, d.G, d.type, d.layout = W, n, 1, 0, _lib.CS_LAYOUT_AOS
    d_state = DeviceBuffer.from_numpy(states if n else np.zeros((W, 1, 13), np.float32))
    d.d_state = d_state.ptr
    d.flags = _lib.CS_OBSTACLES_SHARED
    d_obs = None
    if obstacles is not None:
        obstacles = np.ascontiguousarray(obstacles, dtype=np.float32)
        d.O, d.Smax = obstacles.shape[-4], obstacles.shape[-3]
        if obstacles.ndim == 5:
            d.flags = 0
        d_obs = DeviceBuffer.from_numpy(obstacles)
        d.d_obstacles = d_obs.ptr
    d_pose = DeviceBuffer.from_numpy(np.ascontiguousarray(np.broadcast_to(np.asarray(pose, np.float32), (W, 3))))
    out = DeviceBuffer((W, int(samples)), np.float32)
    check(_lib.load().cs_laser_scan(C.byref(d), C.c_void_p(d_pose.ptr), C.c_int(3), C.c_float(range_), C.c_int(int(samples)),
                                    C.c_float(max_distance), C.c_void_p(out.ptr), C.c_void_p(stream)))
    return out.download(stream)


class LaserSensor:
    """Simulated laser range finder (sensors.py:6-71)."""

    def __init__(self, init_pos, init_yaw: float, range: float, samples: int, max_distance: float, uncertainty=None):
        self.range = range
        self.samples = samples
        if max_distance > 10:
            raise ValueError("Maxium distance for laser is 10 meters")
        self.max_distance = max_distance
        self.uncertainty = uncertainty
        self.update_pose(init_pos, init_yaw)

    def update_pose(self, position, yaw: float):
        if yaw > math.pi or yaw < -math.pi:
            raise ValueError("Angle passed ust be wrapped between [-pi,pi]")
        self.position = position
        self.yaw = yaw

    def get_laser_measurements(self, humans, walls):
        angles = np.linspace(self.yaw - (self.range / 2), self.yaw + (self.range / 2), self.samples)  # not wrapped (:52-53)
        S = np.zeros((1, len(humans), 13), np.float32)
        for i, h in enumerate(humans):
            S[0, i, 0:2] = h.position
            S[0, i, 8] = h.radius
        pose = np.array([self.position[0], self.position[1], self.yaw], dtype=np.float32)
        meas = laser_scan(S, pose, self.range, self.samples, self.max_distance, walls_to_array(walls))[0].astype(PRECISION)
        if self.uncertainty is not None:
            meas = np.array([self.add_uncertainty(m) for m in meas], dtype=PRECISION)
        return dict(zip(angles, meas))

    def add_uncertainty(self, measurement: float):
        measurement = np.random.normal(measurement, self.uncertainty)
        return max(min(measurement, self.max_distance), 0)
