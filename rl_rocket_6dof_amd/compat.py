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

try:  # pragma: no cover
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
    HAVE_SB3 = True
except Exception:  # pragma: no cover
    _SB3VecEnv = None
    HAVE_SB3 = False


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
