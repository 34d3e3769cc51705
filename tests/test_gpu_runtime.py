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
