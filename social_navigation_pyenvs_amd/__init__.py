"""MI355X-native batched crowd stepper behind the Social-Navigation-PyEnvs step()/reset() API.

Layout (only what the hot path needs):
  csrc/            hand-written HIP kernels (gfx950) + the C ABI of include/crowdstep.h
  _lib.py          ctypes binding of libcrowdstep.so -- raises if the library is missing
  batched.py       device-resident worlds (W x N agents) driven through the C ABI
  social_gym/      host-side mirror of the reference's interface for this path
                   (SocialNavGym, MotionModelManager, update_humans_parallel, agents ...)
  crowd_nav/       the CrowdNav state / action tuples the boundary exchanges
"""
__version__ = "0.1.0"
