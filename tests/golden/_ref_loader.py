"""Load the *reference* implementation (read-only, /root/reference) for golden generation.

TEST INFRASTRUCTURE ONLY.  This module is used by ``generate_golden.py`` in the build
container, where /root/reference exists.  Nothing here is imported by the product package,
by ``bench.py`` or by any test at run time (the GPU box has no /root/reference).

The reference needs ``gym`` (0.21) and ``pyvista`` at import time; neither is installed here.
We install tiny in-memory stand-ins that follow gym-0.21 semantics for the only pieces the hot
path touches (``gym.Env.seed/close`` and ``gym.spaces.Box`` low/high/shape/dtype/seed/sample/
contains) and an empty ``pyvista``.  The reference *source files themselves* are executed
unmodified from where they lie.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
import warnings

import numpy as np

REF_ROOT = os.environ.get("R6_REFERENCE_ROOT", "/root/reference")


class _Box:
    """gym-0.21 ``spaces.Box`` semantics (bounded case only)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self.np_random = np.random.RandomState(0)

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        return self.np_random.uniform(low=self.low, high=self.high, size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(
            np.can_cast(x.dtype, self.dtype)
            and x.shape == self.shape
            and np.all(x >= self.low)
            and np.all(x <= self.high)
        )


class _Env:
    metadata: dict = {}

    def seed(self, seed=None):
        return [seed]

    def close(self):
        return None


class _Wrapper(_Env):
    """gym-0.21 ``Wrapper``: forwards unknown attributes to the wrapped env."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kw):
        return self.env.reset(**kw)


class _ObservationWrapper(_Wrapper):
    def reset(self, **kw):
        return self.observation(self.env.reset(**kw))

    def step(self, action):
        o, r, d, i = self.env.step(action)
        return self.observation(o), r, d, i


class _RewardWrapper(_Wrapper):
    def step(self, action):
        o, r, d, i = self.env.step(action)
        return o, self.reward(r), d, i


def _install_stubs():
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = _Box
    gym.Env = _Env
    _Env.unwrapped = property(lambda self: self)
    gym.Wrapper, gym.ObservationWrapper, gym.RewardWrapper = _Wrapper, _ObservationWrapper, _RewardWrapper
    gym.spaces = spaces
    wandb = types.ModuleType("wandb")
    wandb.run = None
    wandb.log = lambda *a, **k: None
    sys.modules.setdefault("wandb", wandb)
    sys.modules.setdefault("gym", gym)
    sys.modules.setdefault("gym.spaces", spaces)
    pv = types.ModuleType("pyvista")
    pv.close_all = lambda: None
    sys.modules.setdefault("pyvista", pv)
    # empty package shells so that `my_environment.utils.simulator` resolves without running
    # my_environment/__init__.py (which needs the gym registry)
    for name in ("my_environment", "my_environment.utils", "my_environment.envs"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = []  # mark as package
            sys.modules[name] = mod


def _load(name: str, relpath: str):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Returns (Simulator6DOF, Rocket6DOF, env_config) from the reference."""
    import yaml

    _install_stubs()
    sim = _load("my_environment.utils.simulator", "my_environment/utils/simulator.py")
    env = _load("my_environment.envs.rocket_env", "my_environment/envs/rocket_env.py")
    with open(os.path.join(REF_ROOT, "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    return sim.Simulator6DOF, env.Rocket6DOF, cfg["env_config"]


def load_reference_wrappers():
    """The reference's reward wrappers (my_environment/wrappers/wrappers.py), executed where they lie."""
    load_reference()
    try:
        import matplotlib  # noqa: F401  (wrappers.py imports pyplot at module top)
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules.setdefault("matplotlib", mpl)
        sys.modules.setdefault("matplotlib.pyplot", mpl.pyplot)
    return _load("my_environment.wrappers.wrappers", "my_environment/wrappers/wrappers.py")
