"""rl_rocket_6dof_amd -- MI355X-native batched Rocket6DOF environment (drop-in for the
``Rocket6DOF.step()`` hot path of Tuxliri/RL_Rocket_6DOF).

Public surface
    BatchedRocket6DOF   N envs per HIP kernel launch, SB3 ``VecEnv``-shaped
    Rocket6DOF          single env with the reference's gym-0.21 class surface (N = 1 view)
    Simulator6DOF       reference ``Simulator6DOF`` surface (dynamics only)
    load_config         reference ``main_6DOF.load_config``
The heavy modules import torch and the HIP library lazily, on first attribute access.
"""
from __future__ import annotations

from .params import default_env_config, load_config, make_params  # noqa: F401

__version__ = "0.1.0"
ENV_ID = "my_environment/Falcon6DOF-v0"   # reference my_environment/__init__.py:4-7

_LAZY = {
    "BatchedRocket6DOF": ("batched_env", "BatchedRocket6DOF"),
    "LazyInfos": ("batched_env", "LazyInfos"),
    "Rocket6DOF": ("rocket_env", "Rocket6DOF"),
    "Simulator6DOF": ("rocket_env", "Simulator6DOF"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(name)


def register():
    """Registers ``my_environment/Falcon6DOF-v0`` -> ``Rocket6DOF`` if a gym registry is importable."""
    from .compat import register_gym_id

    return register_gym_id(ENV_ID, f"{__name__}.rocket_env:Rocket6DOF")


# my_environment/__init__.py:4-7 of the reference registers the id when the package is imported; so does this one
# whenever a gym registry is importable (a no-op otherwise -- gym is not a dependency).
try:
    register()
except Exception:  # pragma: no cover - registration must never break the import
    pass
