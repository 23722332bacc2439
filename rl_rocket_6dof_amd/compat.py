"""gym / SB3 interoperability without a hard dependency.

The reference targets gym 0.21 + stable-baselines3 1.6 (reference setup.py:6, requirements.txt:82).
Neither is installed in the build image, so the env classes derive from whatever is importable
(``gym`` or ``gymnasium``) and otherwise from the minimal stand-ins below, which follow gym-0.21
semantics for the only members the reference uses (``Env.seed/close``, ``spaces.Box``
``low/high/shape/dtype/seed/sample/contains``).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gym as _gym
    from gym import spaces as _spaces
    HAVE_GYM = "gym"
except Exception:  # pragma: no cover
    try:
        import gymnasium as _gym
        from gymnasium import spaces as _spaces
        HAVE_GYM = "gymnasium"
    except Exception:
        _gym = None
        _spaces = None
        HAVE_GYM = None


class _Box:
    """gym-0.21 ``spaces.Box`` (bounded)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self.np_random = np.random.RandomState()

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        return self.np_random.uniform(low=self.low, high=self.high, size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                    and np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class _Env:
    metadata: dict = {}
    reward_range = (-float("inf"), float("inf"))

    @property
    def unwrapped(self):
        return self

    def seed(self, seed=None):
        return [seed]

    def close(self):
        return None


Box = _spaces.Box if _spaces is not None else _Box
Env = _gym.Env if _gym is not None else _Env

class _VecEnv:
    """Stand-in for ``stable_baselines3.common.vec_env.VecEnv`` (SB3 1.6 public API) used when SB3 is not
    importable: the constructor contract ``VecEnv.__init__(num_envs, observation_space, action_space)`` and the
    concrete helpers SB3 code calls on any VecEnv (``step``, ``unwrapped``, ``getattr_depth_check``,
    ``_get_indices``).  The abstract part (reset / step_async / step_wait / close / get_attr / set_attr /
    env_method / env_is_wrapped / seed) is implemented by the subclass."""

    metadata = {"render.modes": []}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self

    def getattr_depth_check(self, name, already_found):
        if hasattr(self, name) and already_found:
            return f"{type(self).__module__}.{type(self).__name__}"
        return None

    def _get_indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices


try:  # pragma: no cover - depends on the environment
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
    HAVE_SB3 = True
except Exception:  # pragma: no cover
    _SB3VecEnv = None
    HAVE_SB3 = False

# Base class of BatchedRocket6DOF: the real SB3 VecEnv whenever SB3 is importable, so that SB3's
# ``isinstance(env, VecEnv)`` checks (BaseAlgorithm._wrap_env) take the env as it is.
VecEnv = _SB3VecEnv if _SB3VecEnv is not None else _VecEnv


def register_gym_id(env_id: str, entry_point: str) -> bool:
    """Registers the reference's id (my_environment/__init__.py:4-7) when a gym registry exists."""
    if _gym is None:
        return False
    try:  # pragma: no cover
        from gym.envs.registration import register
    except Exception:  # pragma: no cover
        try:
            from gymnasium.envs.registration import register
        except Exception:
            return False
    try:  # pragma: no cover
        register(id=env_id, entry_point=entry_point)
        return True
    except Exception:  # already registered, or id rejected by a newer gym
        return False
