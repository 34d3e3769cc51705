"""SURVEY.md §8 row f4 (laser range finder): the numpy oracle against golden G9 captured from the reference
(LaserSensor.get_laser_measurements), and the HIP kernel (cs_laser_scan) against both."""
import os
import sys
import types

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from golden_io import load_cases  # noqa: E402

from oracle import crowd_oracle as orc  # noqa: E402


def test_oracle_matches_golden_g9():
    n_rays = 0
    for c in load_cases("g9_laser"):
        if c["uncertainty"] is not None:
            continue
        obs = c["obstacles"] if c["obstacles"].shape[0] else None
        ang, m = orc.laser_scan(c["pos"], c["yaw"], c["range"], c["samples"], c["max_distance"], c["human_pos"],
                                c["human_radius"], obs)
        np.testing.assert_array_equal(ang, c["angles"])
        np.testing.assert_allclose(m, c["measurements"], rtol=0, atol=1e-12)
        n_rays += c["samples"]
    assert n_rays > 1000


def _near_discontinuity(c, k, obs, eps=2e-5):
    """True when ray k flips between hit and miss (or between targets) under a tiny rotation: f32 may take either side."""
    vals = []
    for d in (-eps, 0.0, eps):
        _, m = orc.laser_scan(c["pos"], c["angles"][k] + d, 0.0, 1, c["max_distance"], c["human_pos"], c["human_radius"], obs)
        vals.append(m[0])
    return max(vals) - min(vals) > 1e-3


@pytest.mark.gpu
def test_kernel_matches_golden_g9_and_host_mirror():
    from social_navigation_pyenvs_amd.social_gym.src.obstacle import Obstacle
    from social_navigation_pyenvs_amd.social_gym.src.sensors import LaserSensor

    checked = flips = 0
    for c in load_cases("g9_laser"):
        n = c["n"]
        humans = [types.SimpleNamespace(position=c["human_pos"][i], radius=float(c["human_radius"][i])) for i in range(n)]
        walls = []
        obs = c["obstacles"] if c["obstacles"].shape[0] else None
        if obs is not None:
            for poly in obs:
                segs = [sg for sg in poly if not np.isnan(sg[0, 0])]
                w = types.SimpleNamespace(segments={j: [list(sg[0]), list(sg[1])] for j, sg in enumerate(segs)})
                walls.append(w)
        laser = LaserSensor(c["pos"].copy(), c["yaw"], c["range"], c["samples"], c["max_distance"], uncertainty=c["uncertainty"])
        np.random.seed(c["noise_seed"])
        out = laser.get_laser_measurements(humans, walls)
        np.testing.assert_array_equal(np.array(list(out.keys())), c["angles"])
        got = np.array(list(out.values()))
        bad = np.flatnonzero(np.abs(got - c["measurements"]) > 2e-5)   # f32 kernel vs the f64 reference, <= 10 m
        for k in bad:
            assert c["uncertainty"] is None and _near_discontinuity(c, k, obs), (k, got[k], c["measurements"][k])
            flips += 1
        checked += c["samples"]
    assert checked > 2000 and flips <= checked // 200
    assert isinstance(Obstacle(None, [[0, 0], [1, 0], [0, 1]]).segments, dict)


@pytest.mark.gpu
def test_batched_scan_on_resident_worlds_equals_oracle():
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 256, 25
    S, goals, P, rb = sc.hybrid_worlds(W, n, "hsfm_farina")
    walls = sc.polygon_walls()
    robot = np.zeros((W, 13), np.float32)
    rng = np.random.default_rng(3)
    robot[:, 0:2] = rng.uniform(-3, 3, (W, 2)); robot[:, 2] = rng.uniform(-np.pi, np.pi, W); robot[:, 8] = 0.3
    for layout in ("aos", "soa"):
        cw = CrowdWorlds(S, goals, P, None, walls, type="hsfm_farina", all_params_equal=True, robot=robot, layout=layout)
        cw.step(0.0125, 20)
        got = cw.laser_scan(np.pi, 61, 8.0)
        assert got.shape == (W, 61) and np.all(got <= 8.0) and np.all(got >= 0)
        St = cw.get_states()
        for w in (0, 17, W - 1):
            _, ref = orc.laser_scan(robot[w, 0:2], robot[w, 2], np.float32(np.pi), 61, 8.0, St[w, :, 0:2], St[w, :, 8], walls,
                                    dtype=np.float64)
            bad = np.abs(got[w] - ref) > 5e-5
            assert bad.sum() <= 1
    with pytest.raises(ValueError):
        cw.laser_scan(np.pi, 61, 12.0)
