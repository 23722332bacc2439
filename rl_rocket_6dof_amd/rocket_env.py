"""Single-environment drop-ins with the reference's class surface, backed by the HIP kernels.

* ``Rocket6DOF``     -- reference my_environment/envs/rocket_env.py:16 (gym 0.21 ``Env``: ``reset() ->
  obs[14]``, ``step(a) -> (obs[14], reward, done, info)``, same constructor keywords as
  config.yaml's ``env_config``, same spaces, same ``info`` keys).
* ``Simulator6DOF``  -- reference my_environment/utils/simulator.py:8 (``step(u) -> (state, status)``,
  ``RHS(t, y, u)``, ``get_thrust_vector_inertial()``, ``states/actions/times`` histories).

Both are N = 1 views of the batched device path (one lane of one wavefront does the work); they
exist so that scripts written against the reference run unchanged, not for speed -- use
``BatchedRocket6DOF`` for throughput.  No CPU implementation: they need the HIP library and a GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .batched_env import BatchedRocket6DOF, LazyInfos, _ptr
from .compat import Box, Env
from .params import (ACTION_NAMES, MAX_GIMBAL, MAX_THRUST, STATE_DIM, STATE_NAMES, R6Params, auto_substeps,
                     default_env_config)

__all__ = ["Rocket6DOF", "Simulator6DOF"]


def _rotmat(q):
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


class Simulator6DOF:
    """simulator.py:8-244 -- 14-state rigid-body rocket, one ``step(u)`` per env timestep."""

    def __init__(self, IC, dt: float = 0.5, *, device=None, integrator: str = "rk4",
                 n_substeps: Optional[int] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("Simulator6DOF needs a ROCm GPU; there is no CPU implementation in this package")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        ic = np.asarray(IC, dtype=np.float64).copy()
        if ic.shape != (STATE_DIM,):
            raise ValueError("IC must have 14 entries")
        self.timestep = dt
        self.t = 0
        self.state = ic
        self.states = [ic]
        self.actions = [[0, 0, 0]]
        self.times = [0]
        # simulator.py:39-67
        self.g0 = 9.81
        self.m = float(ic[13])
        self.length = 40
        self.base_radius = 3.66 / 2
        self.Isp = 360
        self.r_T_B = [-15, 0, 0]
        self.r_cp_B = [5, 0, 0]
        self._p = R6Params()
        self._p.dt = float(dt)
        self._p.integrator = {"euler": 0, "rk4": 1, "rk45": 2}[integrator]
        self._p.n_substeps = int(auto_substeps(dt) if n_substeps is None else n_substeps)
        # One device image [state 14 x ld | epconst 2 x ld | u | status | dy 14 x ld] (rows of ld = 64 floats) next to a pinned host
        # image of the same layout: a step is one upload (state, u), one launch, one download (state, status) and one
        # synchronisation -- the state lives on the host between calls, as in the reference (scripts overwrite `SIM.state`).
        self._ld = ld = 64
        rows = STATE_DIM + 2 + 1 + 1 + STATE_DIM
        self._dev = torch.zeros((rows, ld), dtype=torch.float32, device=self.device)
        self._host = torch.zeros((rows, ld), dtype=torch.float32).pin_memory()
        self._h = self._host.numpy()
        self._state, self._epconst = self._dev[:STATE_DIM], self._dev[STATE_DIM:STATE_DIM + 2]
        self._u = self._dev[STATE_DIM + 2, :3].view(1, 3)
        self._status = self._dev[STATE_DIM + 3].view(torch.uint8)[:1]
        self._dy = self._dev[STATE_DIM + 4:]
        self._h[STATE_DIM, 0] = self.m
        self._dev.copy_(self._host)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _upload(self, state, u):
        h = self._h
        h[:STATE_DIM, 0] = np.asarray(state, dtype=np.float32)
        h[STATE_DIM + 2, :3] = np.asarray(u, dtype=np.float32)
        up = STATE_DIM + 3
        self._dev[:up].copy_(self._host[:up], non_blocking=True)

    def _download(self, lo, hi):
        self._host[lo:hi].copy_(self._dev[lo:hi], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()

    def step(self, u):
        """simulator.py:69-104 -- returns (state[14], status); status 1 = ground event."""
        with torch.cuda.device(self.device):
            self._upload(self.state, u)
            rc = self.lib.r6_sim_step(C.byref(self._p), _ptr(self._state), _ptr(self._epconst), _ptr(self._u),
                                      _ptr(self._status), 1, self._ld, self._stream())
            _lib.check(rc, "r6_sim_step", self.lib)
            self._download(0, STATE_DIM + 4)
        self.state = self._h[:STATE_DIM, 0].astype(np.float64)
        status = int(self._h[STATE_DIM + 3].view(np.uint8)[0])
        self.t = round(self.t + self.timestep, 3)                  # simulator.py:92
        self.times.append(self.t)
        self.states.append(self.state)
        self.actions.append(u)
        return self.state, status

    def RHS(self, t, state, u):
        """simulator.py:106-143."""
        with torch.cuda.device(self.device):
            self._upload(state, u)
            rc = self.lib.r6_rhs(C.byref(self._p), _ptr(self._state), _ptr(self._epconst), _ptr(self._u),
                                 _ptr(self._dy), 1, self._ld, self._stream())
            _lib.check(rc, "r6_rhs", self.lib)
            self._download(STATE_DIM + 4, 2 * STATE_DIM + 4)
        return self._h[STATE_DIM + 4:, 0].astype(np.float64)

    def get_thrust_vector_inertial(self):
        """simulator.py:177-186 -- R(q of states[-1]) @ T_B(actions[-1])  (host-side helper)."""
        u = np.asarray(self.actions[-1], dtype=np.float32)
        cy, cz, sy, sz = np.cos(u[0]), np.cos(u[1]), np.sin(u[0]), np.sin(u[1])
        t_b = np.array([cy * cz, sy * cz, sz], dtype=np.float64) * float(u[2])
        return _rotmat(self.states[-1][6:10]) @ t_b


class _SimView:
    """What wrappers and scripts read through ``env.SIM`` (wrappers.py:74, montecarlo_script.py:35)."""

    def __init__(self, ic, dt):
        self.timestep = dt
        self.t = 0
        self.state = ic
        self.states = [ic]
        self.actions = [[0, 0, 0]]
        self.times = [0]
        self.m = float(ic[13])

    def get_thrust_vector_inertial(self):
        u = np.asarray(self.actions[-1], dtype=np.float32)
        cy, cz, sy, sz = np.cos(u[0]), np.cos(u[1]), np.sin(u[0]), np.sin(u[1])
        return _rotmat(self.states[-1][6:10]) @ (np.array([cy * cz, sy * cz, sz], dtype=np.float64) * float(u[2]))


class Rocket6DOF(Env):
    """rocket_env.py:16 -- single 6DOF rocket-landing env (gym 0.21 API, 4-tuple ``step``)."""

    metadata = {"render.modes": ["human", "rgb_array"], "render_fps": 1 / 0.1}

    def __init__(self, IC=None, ICRange=None, timestep=0.1, seed=42, reward_shaping_type="acceleration",
                 reward_coeff=None, trajectory_limits=None, landing_params=None, *, device=None,
                 integrator: str = "rk4", n_substeps: Optional[int] = None):
        super().__init__()
        base = default_env_config()
        cfg = {
            "IC": base["IC"] if IC is None else IC,
            "ICRange": base["ICRange"] if ICRange is None else ICRange,
            "timestep": timestep,
            "seed": seed,
            "reward_shaping_type": reward_shaping_type,
            "reward_coeff": base["reward_coeff"] if reward_coeff is None else reward_coeff,
            "trajectory_limits": base["trajectory_limits"] if trajectory_limits is None else trajectory_limits,
            "landing_params": base["landing_params"] if landing_params is None else landing_params,
        }
        self.state_names = list(STATE_NAMES)
        self.action_names = list(ACTION_NAMES)
        # host_mirror: all of a step's outputs come back in ONE device->host copy (one env is read back every step)
        self._vec = BatchedRocket6DOF(1, cfg, device=device, integrator=integrator, n_substeps=n_substeps,
                                      auto_reset=False, record_terms=True, record_flags=True, seed=seed, host_mirror=True)
        spec = self._vec.spec
        self.ICMean, self.ICRange = spec.ic_mean, spec.ic_range
        self.timestep = timestep
        self.metadata = dict(self.metadata, render_fps=1 / timestep)
        self.reward_coefficients = spec.reward_coefficients
        self.init_space = Box(low=self.ICMean - self.ICRange / 2, high=self.ICMean + self.ICRange / 2)
        self.max_gimbal, self.max_thrust = MAX_GIMBAL, MAX_THRUST
        self.state_normalizer = spec.state_normalizer
        self.position_bounds_space = Box(low=spec.bounds_low, high=spec.bounds_high, dtype=np.float32)
        self.observation_space = Box(low=-1, high=1, shape=(14,))
        self.action_space = Box(low=-1, high=1, shape=(3,))
        self.attitude_traj_limit = spec.attitude_traj_limit
        self.target_r, self.maximum_v = spec.target_r, spec.maximum_v
        self.landing_target = [0, 0, 0]
        self.landing_attitude_limit = spec.landing_attitude_limit
        self.omega_lim = spec.omega_lim
        self.waypoint = spec.waypoint
        self.shaping_type = spec.shaping_type
        self.state = None
        self.SIM: Optional[_SimView] = None
        self.action = np.array([0.0, 0.0, 0.0])
        self.initial_condition = None

    # rocket_env.py:619-621
    def seed(self, seed: int = 42):
        self._vec.seed(seed)
        return [seed]

    def _begin_episode(self, obs):
        h = self._vec.fetch_host()
        ic = h["state"][:, 0].copy()
        self.initial_condition = ic
        self.state = ic
        self.SIM = _SimView(ic, self.timestep)
        return h["obs"][0].copy()

    def reset(self):
        """rocket_env.py:180-199."""
        return self._begin_episode(self._vec.reset())

    def reset_to(self, initial_condition):
        """Starts an episode from an explicit initial condition (quaternion normalised like
        rocket_env.py:190).  Not in the reference; used for parity tests."""
        ic = np.array(initial_condition, dtype=np.float32)
        ic[6:10] = ic[6:10] / np.linalg.norm(ic[6:10])
        return self._begin_episode(self._vec.set_state(ic[None, :]))

    def _denormalize_action(self, action):
        """rocket_env.py:509-521."""
        return np.float32([action[0] * self.max_gimbal, action[1] * self.max_gimbal,
                           (action[2] + 1) / 2.0 * self.max_thrust])

    def step(self, normalized_action):
        """rocket_env.py:201-231."""
        a = np.asarray(normalized_action, dtype=np.float32).reshape(1, 3)
        self.action = self._denormalize_action(a[0])
        h = self._vec.step_host(a)
        self.state = h["state"][:, 0].astype(np.float64)
        sim = self.SIM
        sim.state = self.state
        sim.t = round(sim.t + self.timestep, 3)
        sim.times.append(sim.t)
        sim.states.append(self.state)
        sim.actions.append(self.action)
        info = LazyInfos.from_host(self._vec, h)[0]
        info.update(state_history=sim.states, action_history=sim.actions, timesteps=sim.times)
        return h["obs"][0].copy(), float(h["reward"][0]), bool(h["done"][0]), info

    def get_state(self):
        return self.state

    def _get_obs(self):
        return (self.state / self.state_normalizer).astype("float32")

    def _get_normalizer(self):
        return self.state_normalizer

    def _check_bounds_violation(self, state):
        return not bool(self.position_bounds_space.contains(np.float32(state[0:3])))

    def used_mass(self):
        return self.SIM.states[0][-1] - self.SIM.states[-1][-1]

    def states_to_dataframe(self):
        import pandas as pd
        return pd.DataFrame(self.SIM.states, columns=self.state_names)

    def actions_to_dataframe(self):
        import pandas as pd
        return pd.DataFrame(self.SIM.actions, columns=self.action_names)

    def render(self, mode: str = "rgb_array"):
        raise NotImplementedError("pyvista rendering is outside the hot path and not provided")

    def close(self) -> None:
        self._vec.close()
        return None
