"""Host-side digestion of the reference's ``env_config`` into the kernel's POD parameter block.

Mirrors ``Rocket6DOF.__init__`` (reference my_environment/envs/rocket_env.py:22-178): same YAML keys
(reference config.yaml:2-63), same derived quantities (state normaliser :90-126, position bounds
:129-134, actuator limits :86-87, attitude / landing limits :159-168).  The ctypes structures mirror
``include/r6dof.h`` field for field; ``_lib`` checks their sizes against the loaded library.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import yaml

STATE_DIM = 14
ACTION_DIM = 3
NUM_TERMS = 8
NUM_STATS = 8
BLOCK = 256
MAX_BLOCKS = 16384

EULER, RK4, RK45 = 0, 1, 2
SHAPING_ACCELERATION, SHAPING_VELOCITY = 0, 1
FLAG_GROUND, FLAG_BOUNDS, FLAG_TRUNCATED = 1, 2, 4

STATE_NAMES = ["x", "y", "z", "vx", "vy", "vz", "q0", "q1", "q2", "q3", "omega1", "omega2", "omega3", "mass"]  # rocket_env.py:52-67
ACTION_NAMES = ["gimbal_y", "gimbal_z", "thrust"]                                                              # rocket_env.py:68
STAT_NAMES = ["episodes", "return_sum", "return_sq_sum", "length_sum", "ground", "bounds", "truncated", "landed"]

MAX_GIMBAL = float(np.deg2rad(20))   # rocket_env.py:86
MAX_THRUST = 981e3                   # rocket_env.py:87
BOUNDS_PENALTY = -50.0               # rocket_env.py:228-229


class R6Params(C.Structure):
    _fields_ = [
        ("dt", C.c_float), ("integrator", C.c_int32), ("n_substeps", C.c_int32), ("shaping", C.c_int32),
        ("max_gimbal", C.c_float), ("max_thrust", C.c_float),
        ("inv_norm", C.c_float * STATE_DIM),
        ("bound_lo", C.c_float * 3), ("bound_hi", C.c_float * 3),
        ("alfa", C.c_float), ("beta", C.c_float), ("eta", C.c_float), ("gamma", C.c_float), ("kappa", C.c_float),
        ("w_r_f", C.c_float), ("w_v_f", C.c_float), ("max_r_f", C.c_float), ("max_v_f", C.c_float),
        ("bounds_penalty", C.c_float),
        ("traj_c0", C.c_float), ("traj_s1", C.c_float), ("traj_c2", C.c_float),
        ("land_c0", C.c_float), ("land_s1", C.c_float), ("land_c2", C.c_float),
        ("land_radius", C.c_float), ("land_vmax", C.c_float), ("omega_lim", C.c_float * 3),
        ("waypoint", C.c_float),
        ("ic_lo", C.c_float * STATE_DIM), ("ic_range", C.c_float * STATE_DIM),
        ("obs_dim", C.c_int32), ("clip_reward", C.c_int32), ("clip_lo", C.c_float), ("clip_hi", C.c_float),
        ("max_episode_steps", C.c_int32), ("auto_reset", C.c_int32),
        ("reward_mode", C.c_int32), ("xi", C.c_float), ("vatt_enable", C.c_int32), ("vatt_threshold", C.c_float),
        ("vatt_weight", C.c_float), ("reserved0", C.c_int32),
    ]


class R6Buffers(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("ld", C.c_int64),
        ("state", C.c_void_p), ("epconst", C.c_void_p), ("action", C.c_void_p), ("obs", C.c_void_p),
        ("reward", C.c_void_p), ("done", C.c_void_p),
        ("flags", C.c_void_p), ("terms", C.c_void_p), ("final_obs", C.c_void_p), ("ep_step", C.c_void_p),
        ("ep_return", C.c_void_p), ("fin_return", C.c_void_p), ("fin_length", C.c_void_p), ("nfev", C.c_void_p), ("stats", C.c_void_p),
        ("step_base", C.c_void_p),
    ]


K_LAST_ONLY = 1


class R6StepK(C.Structure):
    """include/r6dof.h ``R6StepK``: K env steps per launch, element strides of the per-step arrays."""
    _fields_ = [
        ("k", C.c_int32), ("flags", C.c_int32),
        ("action_stride", C.c_int64), ("obs_stride", C.c_int64), ("reward_stride", C.c_int64), ("done_stride", C.c_int64),
        ("flags_stride", C.c_int64), ("nfev_stride", C.c_int64), ("terms_stride", C.c_int64),
        ("final_obs_stride", C.c_int64), ("fin_stride", C.c_int64),
    ]


class R6Rollout(C.Structure):
    """include/r6dof.h ``R6Rollout``: K closed-loop rollout steps per launch; contiguous (k, n, .) arrays."""
    _fields_ = [
        ("k", C.c_int32), ("deterministic", C.c_int32),
        ("obs_seq", C.c_void_p), ("raw_action", C.c_void_p), ("value", C.c_void_p), ("logp", C.c_void_p),
        ("reward", C.c_void_p), ("done", C.c_void_p), ("flags", C.c_void_p), ("boot_value", C.c_void_p),
    ]


class R6PolicyNet(C.Structure):
    """include/r6dof.h ``R6PolicyNet``: device pointers to fp32 weights in torch.nn.Linear layout."""
    _fields_ = [
        ("obs_dim", C.c_int32), ("hidden1", C.c_int32), ("hidden2", C.c_int32), ("shared", C.c_int32),
        ("pi_w1", C.c_void_p), ("pi_b1", C.c_void_p), ("pi_w2", C.c_void_p), ("pi_b2", C.c_void_p),
        ("pi_w3", C.c_void_p), ("pi_b3", C.c_void_p),
        ("vf_w1", C.c_void_p), ("vf_b1", C.c_void_p), ("vf_w2", C.c_void_p), ("vf_b2", C.c_void_p),
        ("vf_w3", C.c_void_p), ("vf_b3", C.c_void_p),
        ("log_std", C.c_void_p),
    ]


# Built-in defaults: the parameter set the reference trains with (the values of its config.yaml; any
# YAML file with the same keys can be passed to load_config instead).
_DEFAULT_ENV_CONFIG = {
    "timestep": 0.1,
    "seed": 42,
    "reward_shaping_type": "acceleration",
    #        x      y      z   vx    vy  vz   q0    q1 q2  q3   w1 w2 w3   mass
    "IC": [2000, -1600, 0, -90, 180, 0, 0.866, 0, 0, -0.5, 0, 0, 0, 41.0e3],
    "ICRange": [10, 200, 0, 30, 30, 0, 0.1, 0.1, 0.1, 0.1, 0.05, 0.05, 0.05, 1.0e3],
    "trajectory_limits": {"attitude_limit": [85, 85, 360]},
    "landing_params": {"waypoint": -1, "landing_radius": 30, "maximum_velocity": 10,
                       "landing_attitude_limit": [10, 10, 360],   # [yaw, pitch, roll] deg, rotation order zyx
                       "omega_lim": [0.2, 0.2, 0.2]},
    "reward_coeff": {"alfa": -0.01, "beta": -1.0e-8, "gamma": -10, "delta": -5, "eta": 0.1, "kappa": 10, "xi": 0.004,
                     "w_r_f": 0, "w_v_f": 0.75,     # 0 disables the corresponding terminal reward
                     "max_r_f": 100, "max_v_f": 50},
}
_DEFAULT_SB3_CONFIG = {"env_id": "my_environment/Falcon6DOF-v0", "policy_type": "MlpPolicy",
                       "total_timesteps": 20.0e6, "max_time": 150}


def load_config(path: Optional[str] = None):
    """``load_config()`` of the reference (main_6DOF.py:18-27): returns (sb3_config, env_config) from a
    YAML file with the reference's layout, or the built-in defaults when no path is given."""
    if path is None:
        import copy
        return copy.deepcopy(_DEFAULT_SB3_CONFIG), copy.deepcopy(_DEFAULT_ENV_CONFIG)
    with open(path) as f:
        cfg = yaml.safe_load(f)
    return cfg["sb3_config"], cfg["env_config"]


def default_env_config() -> dict:
    return load_config()[1]


@dataclass
class EnvSpec:
    """Python-visible derived quantities (what the reference keeps as attributes of the env)."""
    ic_mean: np.ndarray          # float32, rocket_env.py:71
    ic_range: np.ndarray         # float32, rocket_env.py:72
    timestep: float
    state_normalizer: np.ndarray  # float64, rocket_env.py:106-126
    bounds_low: np.ndarray       # float32, rocket_env.py:129-134
    bounds_high: np.ndarray
    reward_coefficients: dict
    attitude_traj_limit: np.ndarray
    landing_attitude_limit: np.ndarray
    target_r: float
    maximum_v: float
    omega_lim: np.ndarray
    waypoint: float
    shaping_type: str
    seed: int


def derive_spec(env_config: dict) -> EnvSpec:
    if "waypoint" not in env_config.get("landing_params", {}):
        # the reference raises KeyError('waypoint') for its own class defaults (rocket_env.py:168)
        raise KeyError("waypoint")
    icm = np.float32(env_config["IC"])
    icr = np.float32(env_config["ICRange"])
    if icm.shape != (STATE_DIM,) or icr.shape != (STATE_DIM,):
        raise ValueError("IC and ICRange must have 14 entries")
    # rocket_env.py:90-104.  ICMean is float32, so with numpy >= 2 scalar promotion t_free_fall and
    # v_max are float32 quantities (that is what the golden vectors were generated with).
    f32 = np.float32
    t_free_fall = (-icm[3] + np.sqrt(icm[3] ** 2 + f32(2 * 9.81) * icm[0])) / f32(9.81)
    inertia, lever_arm = 6.04e6, 15.0
    omega_max = MAX_THRUST * math.sin(MAX_GIMBAL) * lever_arm / inertia * float(t_free_fall) / 5.0
    v_max = float(f32(2 * 9.81) * t_free_fall)
    normalizer = np.maximum(
        np.array([float(f32(1.2) * abs(icm[0])), float(f32(1.5) * abs(icm[1])), float(f32(1.5) * abs(icm[2])),   # float32 products, like t_free_fall
                  v_max, v_max, v_max, 1.1, 1.1, 1.1, 1.1, omega_max, omega_max, omega_max,
                  float(icm[13] + icr[13])]), 1.0)
    hi = (0.9 * np.maximum(normalizer[0:3], 200)).astype(np.float32)
    lo = np.concatenate([[-30.0], -0.9 * np.maximum(normalizer[1:3], 200)]).astype(np.float32)
    shaping = env_config.get("reward_shaping_type", "acceleration")
    if shaping not in ("acceleration", "velocity"):
        raise ValueError(f"unknown reward_shaping_type {shaping!r}")
    return EnvSpec(
        ic_mean=icm, ic_range=icr, timestep=float(env_config.get("timestep", 0.1)),
        state_normalizer=normalizer, bounds_low=lo, bounds_high=hi,
        reward_coefficients=dict(env_config["reward_coeff"]),
        attitude_traj_limit=np.deg2rad(env_config["trajectory_limits"]["attitude_limit"]),
        landing_attitude_limit=np.deg2rad(env_config["landing_params"]["landing_attitude_limit"]),
        target_r=float(env_config["landing_params"]["landing_radius"]),
        maximum_v=float(env_config["landing_params"]["maximum_velocity"]),
        omega_lim=np.array([0.2, 0.2, 0.2]),  # rocket_env.py:167: hard-coded, the config value is ignored
        waypoint=float(env_config["landing_params"]["waypoint"]),
        shaping_type=shaping, seed=int(env_config.get("seed", 42)),
    )


def _cos_limit(L: float) -> float:
    """|atan2| limit -> cos threshold (see include/r6dof.h); L >= pi can never be exceeded."""
    return math.cos(L) if 0.0 <= L < math.pi else -2.0


def _sin_limit(L: float) -> float:
    return math.sin(L) if 0.0 <= L < math.pi / 2 else 2.0


def auto_substeps(dt: float) -> int:
    """Fixed RK4 tracks the reference's adaptive RK45 to < 4e-7 per step for h <= 0.1 s;
    longer env steps are split into equal sub-steps of at most 0.1 s."""
    return max(1, int(math.ceil(dt / 0.1 - 1e-9)))


def make_params(env_config: dict, *, integrator: str = "rk4", n_substeps: Optional[int] = None,
                timestep: Optional[float] = None, remove_mass: bool = False, clip_reward=None,
                max_episode_steps: int = 0, auto_reset: bool = True, reward_annealing: bool = False,
                vertical_attitude_reward=None):
    """Returns (R6Params, EnvSpec).  ``reward_annealing``: RewardAnnealing wrapper (wrappers.py:39-61);
    ``vertical_attitude_reward``: None or (threshold_height, weight) of VerticalAttitudeReward (:128-155)."""
    spec = derive_spec(env_config)
    dt = float(spec.timestep if timestep is None else timestep)
    p = R6Params()
    p.dt = dt
    p.integrator = {"euler": EULER, "rk4": RK4, "rk45": RK45}[integrator]
    p.n_substeps = int(auto_substeps(dt) if n_substeps is None else n_substeps)
    p.shaping = SHAPING_ACCELERATION if spec.shaping_type == "acceleration" else SHAPING_VELOCITY
    p.max_gimbal = MAX_GIMBAL
    p.max_thrust = MAX_THRUST
    inv = 1.0 / spec.state_normalizer
    for j in range(STATE_DIM):
        p.inv_norm[j] = inv[j]
        p.ic_lo[j] = float(spec.ic_mean[j] - spec.ic_range[j] / np.float32(2))   # rocket_env.py:78-81 (float32)
        p.ic_range[j] = float(spec.ic_range[j])
    for j in range(3):
        p.bound_lo[j] = float(spec.bounds_low[j])
        p.bound_hi[j] = float(spec.bounds_high[j])
        p.omega_lim[j] = float(spec.omega_lim[j])
    c = spec.reward_coefficients
    p.alfa, p.beta, p.eta, p.gamma, p.kappa = c["alfa"], c["beta"], c["eta"], c["gamma"], c["kappa"]
    p.w_r_f, p.w_v_f, p.max_r_f, p.max_v_f = c["w_r_f"], c["w_v_f"], c["max_r_f"], c["max_v_f"]
    p.bounds_penalty = BOUNDS_PENALTY
    t, l = spec.attitude_traj_limit, spec.landing_attitude_limit
    p.traj_c0, p.traj_s1, p.traj_c2 = _cos_limit(t[0]), _sin_limit(t[1]), _cos_limit(t[2])
    p.land_c0, p.land_s1, p.land_c2 = _cos_limit(l[0]), _sin_limit(l[1]), _cos_limit(l[2])
    p.land_radius = spec.target_r
    p.land_vmax = spec.maximum_v
    p.waypoint = spec.waypoint
    p.obs_dim = 13 if remove_mass else 14
    if clip_reward is None:
        p.clip_reward, p.clip_lo, p.clip_hi = 0, 0.0, 0.0
    else:
        p.clip_reward, p.clip_lo, p.clip_hi = 1, float(clip_reward[0]), float(clip_reward[1])
    p.max_episode_steps = int(max_episode_steps)
    p.auto_reset = 1 if auto_reset else 0
    p.reward_mode = 1 if reward_annealing else 0
    p.xi = float(c.get("xi", 0.01))                       # wrappers.py:42
    if vertical_attitude_reward is None:
        p.vatt_enable, p.vatt_threshold, p.vatt_weight = 0, 1e-3, -0.5
    else:
        p.vatt_enable, p.vatt_threshold, p.vatt_weight = 1, float(vertical_attitude_reward[0]), float(vertical_attitude_reward[1])
    return p, spec


class R6PolicyGrad(C.Structure):
    """include/r6dof.h ``R6PolicyGrad``: device pointers to the gradient buffers (vf_w1..vf_b2: separate value MLP only)."""
    _fields_ = [(k, C.c_void_p) for k in ("w1", "b1", "w2", "b2", "pi_w3", "pi_b3", "vf_w3", "vf_b3", "log_std",
                                          "vf_w1", "vf_b1", "vf_w2", "vf_b2")]
