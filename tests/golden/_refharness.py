"""Harness that makes the *reference* importable in the build container.

TEST INFRASTRUCTURE ONLY.  Used by ``make_golden.py`` (fixture generation) and by the
optional ``-m "not gpu"`` cross-checks that run only when ``/root/reference`` exists.
Nothing here is imported by the shipped package; nothing from the reference is copied:
the stand-ins below only satisfy ``import`` statements of packages that are missing in
this image (numba, pygame, gymnasium, rvo2, socialforce) with inert objects.

Recipe follows SURVEY.md Appendix D.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CROWDSTEP_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "social_gym"))


def _identity_jit(*args, **kwargs):
    # supports both ``@njit`` and ``@njit(nogil=True, ...)``
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


class _Anything:
    """Object that absorbs any attribute access / call (for rendering-only APIs)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()

    def __iter__(self):
        return iter(())

    def __len__(self):
        return 0

    def get_size(self):
        return (1, 1)

    def get_rect(self, **k):
        r = _Anything()
        r.__dict__.update(x=0, y=0, centerx=0, centery=0)
        return r

    def get_ticks(self):
        return 0

    def get_fps(self):
        return 0.0


class _Sprite:
    def __init__(self, *a, **k):
        pass


class _Group:
    def __init__(self, *a):
        self._items = list(a)

    def add(self, *items):
        self._items.extend(items)

    def empty(self):
        self._items.clear()

    def sprites(self):
        return list(self._items)

    def __len__(self):
        return len(self._items)

    def __iter__(self):
        return iter(self._items)


def _install_stub_modules() -> None:
    # numba -------------------------------------------------------------
    if "numba" not in sys.modules:
        numba = types.ModuleType("numba")
        numba.njit = _identity_jit
        numba.jit = _identity_jit
        numba.prange = range
        sys.modules["numba"] = numba
    # pygame ------------------------------------------------------------
    if "pygame" not in sys.modules:
        pg = types.ModuleType("pygame")
        pg.init = lambda *a, **k: None
        pg.quit = lambda *a, **k: None
        pg.SRCALPHA = 0
        pg.Surface = _Anything
        pg.draw = _Anything()
        pg.display = _Anything()
        pg.key = _Anything()
        pg.event = _Anything()
        pg.font = _Anything()
        pg.transform = _Anything()
        pg.time = _Anything()
        pg.image = _Anything()
        pg.mouse = _Anything()
        sprite = types.ModuleType("pygame.sprite")
        sprite.Sprite = _Sprite
        sprite.Group = _Group
        pg.sprite = sprite

        def _pg_getattr(name):  # K_* constants, QUIT, ...
            if name.startswith("__"):
                raise AttributeError(name)
            return 0

        pg.__getattr__ = _pg_getattr
        sys.modules["pygame"] = pg
        sys.modules["pygame.sprite"] = sprite
    # rvo2 / socialforce --------------------------------------------------
    if "rvo2" not in sys.modules:
        rvo2 = types.ModuleType("rvo2")

        class PyRVOSimulator:  # noqa: D401 - the real library is absent
            def __init__(self, *a, **k):
                raise RuntimeError("rvo2 is not available in this container")

        rvo2.PyRVOSimulator = PyRVOSimulator
        sys.modules["rvo2"] = rvo2
    if "socialforce" not in sys.modules:
        sf = types.ModuleType("socialforce")
        sf.Simulator = _Anything
        sys.modules["socialforce"] = sf
    # gymnasium -----------------------------------------------------------
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")

        class Env:
            pass

        gym.Env = Env
        spaces = types.ModuleType("gymnasium.spaces")
        spaces.Discrete = lambda n: ("Discrete", n)
        gym.spaces = spaces
        envs = types.ModuleType("gymnasium.envs")
        reg = types.ModuleType("gymnasium.envs.registration")
        reg.register = lambda **k: None
        envs.registration = reg
        gym.envs = envs
        gym.make = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces
        sys.modules["gymnasium.envs"] = envs
        sys.modules["gymnasium.envs.registration"] = reg


def import_reference():
    """Import the reference packages; returns a namespace of the modules we probe."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    import numpy as np

    sys.dont_write_bytecode = True
    if not hasattr(np, "NaN"):
        np.NaN = np.nan  # reference pins numpy 1.26 (motion_model_manager.py:264)
    import torch  # noqa: F401  (import before the stubs: torch inspects sys.modules)

    _install_stub_modules()
    for m in ("numba", "pygame", "pygame.sprite", "rvo2", "socialforce", "gymnasium",
              "gymnasium.spaces", "gymnasium.envs", "gymnasium.envs.registration"):
        if not hasattr(sys.modules[m], "__file__"):
            sys.modules[m].__file__ = f"<crowdstep-stub:{m}>"
    for p in (REFERENCE_ROOT, os.path.join(REFERENCE_ROOT, "crowd_nav")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import logging

    logging.disable(logging.INFO)
    ns = types.SimpleNamespace()
    import social_gym.src.forces_parallel as fp
    import social_gym.src.motion_model_manager as mmm
    import social_gym.social_nav_sim as sim
    import social_gym.social_nav_gym as gymmod
    import social_gym.src.utils as utils
    import social_gym.src.robot_agent as robot_agent
    import crowd_nav.utils.action as action
    import crowd_nav.utils.state as state

    ns.fp, ns.mmm, ns.sim, ns.gym, ns.utils = fp, mmm, sim, gymmod, utils
    ns.robot_agent, ns.action, ns.state = robot_agent, action, state
    return ns
