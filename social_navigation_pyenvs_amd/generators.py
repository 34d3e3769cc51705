"""Device-side scenario generators: ``SocialNavGym.reset`` for W worlds in one launch.

Host mirror of ``cs_generate_worlds`` (include/crowdstep.h): the reference seeds numpy's legacy global stream with
``offset[phase] + case`` and runs one of three rejection-sampling generators per world
(/root/reference/social_gym/social_nav_gym.py:135-197, social_gym/social_nav_sim.py:200-431); the HIP kernel
restates that stream (MT19937, 53-bit doubles) and the generators draw for draw, one lane per world, and writes the
state / goal / robot rows straight into the device buffers of a ``CrowdWorlds``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DeviceBuffer, check

SCENARIOS = {"circle_crossing": 0, "parallel_traffic": 1, "circular_crossing_with_static_obstacles": 2,
             "hybrid_scenario": 3}
SCENARIO_NAMES = {v: k for k, v in SCENARIOS.items()}
MAX_PLACEMENT_TRIES = 100000  # same bound as the host generators (social_gym/social_nav_sim.py)


class cs_generator(C.Structure):
    _fields_ = [
        ("scenario", C.c_int32), ("n", C.c_int32), ("insert_robot", C.c_int32), ("randomize_attributes", C.c_int32),
        ("randomize_positions", C.c_int32), ("max_tries", C.c_int32),
        ("circle_radius", C.c_double), ("traffic_length", C.c_double), ("traffic_height", C.c_double),
        ("robot_radius", C.c_double), ("human_mass", C.c_double), ("robot_mass", C.c_double),
        ("robot_desired_speed", C.c_double),
    ]


def phase_seeds(phase: str, first_case: int, W: int, case_capacity=None) -> np.ndarray:
    """Seeds ``offset[phase] + test_case`` of the W consecutive cases ``first_case + w``, i.e. what
    ``reset(phase, test_case=first_case + w)`` seeds with (social_nav_gym.py:74, 126, 135-137; the modulo by
    ``case_size`` only applies to the counter's own increment, :197)."""
    cap = case_capacity or {"train": np.iinfo(np.uint32).max - 2000, "val": 1000, "test": 1000}
    offset = {"train": cap["val"] + cap["test"], "val": 0, "test": cap["val"]}[phase]
    return (offset + int(first_case) + np.arange(W, dtype=np.int64)).astype(np.uint32)


def make_generator(cw, scenario, *, insert_robot=True, randomize_attributes=False, randomize_positions=True, circle_radius=7,
                   traffic_length=14, traffic_height=3, robot_radius=0.3, human_mass=75, robot_mass=80, robot_desired_speed=1,
                   max_tries=MAX_PLACEMENT_TRIES) -> cs_generator:
    if isinstance(scenario, str):
        scenario = SCENARIOS[scenario]
    g = cs_generator()
    g.scenario, g.n = int(scenario), int(cw.n)
    g.insert_robot, g.randomize_attributes, g.randomize_positions = int(insert_robot), int(randomize_attributes), int(randomize_positions)
    g.max_tries = int(max_tries)
    g.circle_radius, g.traffic_length, g.traffic_height = float(circle_radius), float(traffic_length), float(traffic_height)
    g.robot_radius, g.human_mass = float(robot_radius), float(human_mass)
    g.robot_mass, g.robot_desired_speed = float(robot_mass), float(robot_desired_speed)
    return g


def generate_worlds_device(cw, gen: cs_generator, d_seeds, d_mask=None, d_status=None, d_scenario=None) -> None:
    """Asynchronous form for a device-resident loop: seeds / mask / status are device pointers (DeviceBuffer, torch CUDA
    tensor or int); nothing is copied to or from the host and the call does not synchronise."""
    from .batched import _ptr

    lib = _lib.load()
    nbytes = int(lib.cs_generate_scratch_bytes(C.c_int(cw.W)))
    d_scratch = cw._buffer("gen_mt19937", (nbytes // 4,), np.uint32)
    d = cw.descriptor()
    check(lib.cs_generate_worlds(C.byref(gen), C.byref(d), C.c_void_p(_ptr(d_seeds)), C.c_void_p(_ptr(d_mask)),
                                 C.c_void_p(_ptr(d_status)), C.c_void_p(_ptr(d_scenario)), C.c_void_p(d_scratch.ptr),
                                 C.c_void_p(cw.stream)))


def generate_worlds(cw, scenario, seeds, *, mask=None, insert_robot=True, randomize_attributes=False,
                    randomize_positions=True, circle_radius=7, traffic_length=14, traffic_height=3, robot_radius=0.3,
                    human_mass=75, robot_mass=80, robot_desired_speed=1, max_tries=MAX_PLACEMENT_TRIES,
                    raise_on_failure=True):
    """Regenerate the worlds of ``cw`` (all, or those where ``mask`` is non-zero) on the device.

    Returns ``(status [W] int32, scenario [W] int32)``; with ``raise_on_failure`` a non-zero status raises what the
    host generators raise (RuntimeError: could not place, ValueError: traffic too dense)."""
    if isinstance(scenario, str):
        scenario = SCENARIOS[scenario]
    g = cs_generator()
    g.scenario, g.n = int(scenario), int(cw.n)
    g.insert_robot, g.randomize_attributes, g.randomize_positions = int(insert_robot), int(randomize_attributes), int(randomize_positions)
    g.max_tries = int(max_tries)
    g.circle_radius, g.traffic_length, g.traffic_height = float(circle_radius), float(traffic_length), float(traffic_height)
    g.robot_radius, g.human_mass = float(robot_radius), float(human_mass)
    g.robot_mass, g.robot_desired_speed = float(robot_mass), float(robot_desired_speed)
    seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint32), (cw.W,)))
    lib = _lib.load()
    d_seeds = cw._upload("gen_seeds", seeds, np.uint32)
    d_mask = None if mask is None else cw._upload("gen_mask", np.broadcast_to(np.asarray(mask).astype(np.int32), (cw.W,)), np.int32)
    d_status = cw._buffer("gen_status", (cw.W,), np.int32)
    d_scn = cw._buffer("gen_scenario", (cw.W,), np.int32)
    nbytes = int(lib.cs_generate_scratch_bytes(C.c_int(cw.W)))
    d_scratch = cw._buffer("gen_mt19937", (nbytes // 4,), np.uint32)
    if mask is not None:
        check(lib.cs_memset(C.c_void_p(d_status.ptr), C.c_int(0), C.c_size_t(cw.W * 4), C.c_void_p(cw.stream)))
        check(lib.cs_memset(C.c_void_p(d_scn.ptr), C.c_int(0xFF), C.c_size_t(cw.W * 4), C.c_void_p(cw.stream)))
    d = cw.descriptor()
    check(lib.cs_generate_worlds(C.byref(g), C.byref(d), C.c_void_p(d_seeds.ptr), C.c_void_p(None if d_mask is None else d_mask.ptr),
                                 C.c_void_p(d_status.ptr), C.c_void_p(d_scn.ptr), C.c_void_p(d_scratch.ptr), C.c_void_p(cw.stream)))
    status = d_status.download(cw.stream)
    scn = d_scn.download(cw.stream)
    if raise_on_failure and np.any(status != 0):
        bad = int(np.flatnonzero(status)[0])
        if status[bad] == 2:
            raise ValueError("Number of humans specified is too big for desided traffic height and length")
        raise RuntimeError(f"world {bad} (seed {int(seeds[bad])}): could not place all humans within {max_tries} tries")
    return status, scn


def static_obstacle_crossing(W, n, model="hsfm_farina", *, first_world=0, radius=14.0, n_static=3, walls=True, layout="soa",
                             stream=None):
    """BASELINE.json configs[4] as bench.py measures it and tests/test_gpu_fullsize.py checks it (ONE function, so both step the
    same worlds): the reference's circular-crossing generator (social_nav_sim.py:200-299: rejection sampling against the placed
    humans AND their goals) run on the device by cs_generate_worlds with seed 1000 + GLOBAL world id (a world does not depend on
    the shard it is generated in), on a circle of `radius`; then the first `n_static` humans are made immobile the way
    circular_crossing_with_static_obstacles builds them (:381-387, 416-417: desired speed 0, radius 0.8, both goals = own
    position, standing on the inner circle radius - 3) -- that generator itself does not terminate beyond ~10 humans -- and
    three shared polygon walls are added (scenarios.polygon_walls()).  Returns the CrowdWorlds batch."""
    from . import scenarios as sc
    from .batched import CrowdWorlds

    obstacles = sc.polygon_walls() if walls else None
    P = np.tile(sc.default_params(model), (n, 1))
    cw = CrowdWorlds(np.zeros((W, n, 13), np.float32), np.full((W, n, 2, 2), np.nan, np.float32), P, None, obstacles, type=model,
                     all_params_equal=True, layout=layout, stream=stream)
    generate_worlds(cw, "circle_crossing", 1000 + int(first_world) + np.arange(W), insert_robot=False, circle_radius=radius)
    k = int(n_static)
    if k > 0:
        S, goals = cw.get_states(), cw.get_goals()
        ang = 2.0 * np.pi * (np.arange(k) + 0.25) / k
        p = (radius - 3.0) * np.stack([np.cos(ang), np.sin(ang)], -1).astype(np.float32)
        S[:, :k, 0:2] = p; S[:, :k, 10:12] = p; S[:, :k, 3:8] = 0.0
        S[:, :k, 8] = 0.8; S[:, :k, 12] = 0.0
        goals[:, :k, 0] = p; goals[:, :k, 1] = p
        cw.set_states(S); cw.set_goals(goals)
    return cw
