"""GPU: one HIP runtime per process whatever the import order (crowdstep first, torch second), checked in a fresh child."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path[:0] = [%(root)r, %(root)r + "/tests", %(root)r + "/tests/golden"]
assert "torch" not in sys.modules
import numpy as np
from social_navigation_pyenvs_amd import _lib
from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym
from test_facade_cpu import make_config
W = 64
env = BatchedSocialNavGym(make_config("hsfm_farina", "hybrid_scenario", 5, False), W)
env.reset(phase="test", first_case=0, device=True)          # crowdstep objects FIRST: hipMalloc, kernels
assert "torch" not in sys.modules, "crowdstep must not need torch for this"
import torch                                                 # torch SECOND
a = torch.zeros((W, 2), dtype=torch.float32, device="cuda")
a[:, 1] = 0.5
for _ in range(3):
    obs, rew, term, trunc, info = env.step_device(a)
torch.cuda.synchronize()
assert obs.is_cuda and obs.shape == (W, 5, 5) and bool(torch.isfinite(obs).all())
hip = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
assert len(hip) == 1, hip                                    # ONE HIP runtime mapped
print("OK", hip[0], _lib.hip_runtime_path)
'''


def test_crowdstep_first_then_torch_share_one_hip_runtime():
    env = dict(os.environ)
    env.pop("CROWDSTEP_HIP_RUNTIME", None)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.strip().splitlines()[-1].startswith("OK")


def test_bench_launches_its_own_ranks_and_prints_one_parsable_line():
    """`python bench.py --gpus 2` with no launcher around it (the rehearsal form for a one-GPU box: both ranks on GPU 0, gloo for the
    barrier): the parent starts two fresh rank processes before touching HIP, relays rank 0's line -- ONE line on stdout, strict JSON
    under 4 KB, n_gpus = 2, one record per rank with its device and PCI address, value = the worlds of BOTH ranks over the slowest
    rank's time -- and `--gpus 2` on a box that shows one GPU (no --same-device) refuses instead of measuring one rank."""
    import json

    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--dist-backend", "gloo", "--steps", "4", "--warmup", "2",
           "--repeats", "3", "--worlds", "512", "--no-other-configs", "--no-cpu-baseline", "--full-json", os.path.join(ROOT, "gpurun_out", "bench_full_test.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist"]["world_size"] == 2 and d["dist"]["backend"] == "gloo" and d["dist"]["launcher"] == "self"
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and all(x["pci"] and x["worlds"] == 512 for x in d["ranks"])
    assert d["config"]["worlds_total"] == 1024 and d["scaling"] == "weak"
    assert abs(d["value"] - 1024 * 25 * 20 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1.5
    from social_navigation_pyenvs_amd import _lib
    if _lib.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                           timeout=300, env=env)
        assert r.returncode == 3 and r.stdout.strip() == "" and "needs 2 visible" in r.stderr


WG_CHILD = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
from social_navigation_pyenvs_amd import scenarios as sc
from social_navigation_pyenvs_amd.batched import CrowdWorlds
out = []
for model, W, n in (("hsfm_farina", 37, 25), ("sfm_helbing", 45, 10), ("hsfm_new_moussaid", 9, 17)):
    S, goals, P, rb = sc.hybrid_worlds(W, n, model)
    cw = CrowdWorlds(S, goals, P, None, None, type=model, all_params_equal=True, respawn_bounds=rb,
                     respawn_worlds=(np.arange(W) %% 2 == 1).astype(np.int32), layout="soa")
    for _ in range(6):
        cw.step(0.0125, 20)
    out.append(hashlib.sha256(np.ascontiguousarray(cw.get_states()).tobytes() + np.ascontiguousarray(cw.get_goals()).tobytes()).hexdigest())
print("WGHASH " + " ".join(out))
"""


def test_one_wavefront_blocks_step_the_same_bits_however_many_share_a_workgroup():
    """The one-wavefront step kernels (k_sfm_step MAXT = 64, the DPP-row kernel) go to the dispatcher four independent wavefronts to a
    workgroup (sfmstep_kernel.h; CROWDSTEP_WG_WAVES = 1 / 2 / 4 is read once per process): the stepped worlds are the same bits whatever the
    width -- grids that are not a multiple of it included (37 x 25: 19 blocks; 45 x 10: 12 DPP-row wavefronts; 9 x 17: 3 blocks)."""
    hashes = {}
    for wg in ("1", "2", "4"):
        env = dict(os.environ, CROWDSTEP_WG_WAVES=wg)
        r = subprocess.run([sys.executable, "-c", WG_CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("WGHASH ")]
        assert line, r.stdout[-500:]
        hashes[wg] = line[-1]
    assert hashes["1"] == hashes["2"] == hashes["4"], hashes
