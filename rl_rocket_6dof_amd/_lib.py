"""ctypes binding of ``csrc/libr6dof.so`` (the C ABI of include/r6dof.h).

There is NO CPU fallback: if the HIP library is missing or does not match the header this module
raises, and every product entry point goes through it."""
from __future__ import annotations

import ctypes as C
import os

from .params import R6Buffers, R6Params, R6PolicyGrad, R6PolicyNet, R6Rollout, R6StepK

ABI_VERSION = 7
_LIBS = {}

SYMBOLS = ("r6_abi_version", "r6_params_size", "r6_buffers_size", "r6_last_error", "r6_step", "r6_step_k", "r6_stepk_size", "r6_reset",
           "r6_observe", "r6_sim_step", "r6_rhs", "r6_policy_net_size", "r6_policy_act", "r6_ppo_loss_grad",
           "r6_policy_grad_size", "r6_ppo_scratch_floats", "r6_ppo_minibatch_grad", "r6_minibatch_adv_norm", "r6_adam_clip_step",
           "r6_gae", "r6_permutation", "r6_tune", "r6_rollout", "r6_rollout_size", "r6_device_status", "r6_clear_device_error")


class R6Error(RuntimeError):
    pass


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libr6dof.so")


def load(path: str | None = None):
    """Loads (once per path) and returns the ctypes handle.  ``path`` defaults to the in-tree
    library; other paths are for A/B-ing kernel builds in one process (scripts/kbench.py)."""
    path = os.path.abspath(path or os.environ.get("R6_LIB", lib_path()))
    if path in _LIBS:
        return _LIBS[path]
    # torch first: its bundled libamdhip64.so (SONAME libamdhip64.so.7) must be the one HIP runtime
    # of the process, so that torch's streams / allocations are valid in our launches.
    import torch  # noqa: F401

    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built. Run `python -m rl_rocket_6dof_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    lib = C.CDLL(path)   # RTLD_LOCAL: several builds can coexist in one process (A/B benchmarking)
    u64, i64, vp = C.c_uint64, C.c_int64, C.c_void_p
    lib.r6_abi_version.restype = C.c_int
    lib.r6_params_size.restype = C.c_size_t
    lib.r6_buffers_size.restype = C.c_size_t
    lib.r6_last_error.restype = C.c_char_p
    lib.r6_step.argtypes = [C.POINTER(R6Params), C.POINTER(R6Buffers), u64, u64, u64, vp]
    lib.r6_stepk_size.restype = C.c_size_t
    lib.r6_step_k.argtypes = [C.POINTER(R6Params), C.POINTER(R6Buffers), C.POINTER(R6StepK), u64, u64, u64, vp]
    lib.r6_reset.argtypes = [C.POINTER(R6Params), C.POINTER(R6Buffers), vp, u64, u64, u64, vp]
    lib.r6_observe.argtypes = [C.POINTER(R6Params), C.POINTER(R6Buffers), vp]
    lib.r6_sim_step.argtypes = [C.POINTER(R6Params), vp, vp, vp, vp, i64, i64, vp]
    lib.r6_rhs.argtypes = [C.POINTER(R6Params), vp, vp, vp, vp, i64, i64, vp]
    lib.r6_policy_net_size.restype = C.c_size_t
    lib.r6_policy_act.argtypes = [C.POINTER(R6PolicyNet), vp, i64, vp, vp, vp, vp, u64, u64, vp, u64, C.c_int, vp, C.c_int, vp]
    f32 = C.c_float
    lib.r6_ppo_loss_grad.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, vp, vp, vp, vp, vp]
    lib.r6_policy_grad_size.restype = C.c_size_t
    lib.r6_ppo_scratch_floats.restype = C.c_size_t
    lib.r6_ppo_minibatch_grad.argtypes = [C.POINTER(R6PolicyNet), vp, vp, vp, vp, vp, vp, i64, i64, vp, f32, f32, f32,
                                          C.POINTER(R6PolicyGrad), vp, vp, C.c_size_t, vp]
    lib.r6_ppo_minibatch_grad.restype = C.c_int
    lib.r6_minibatch_adv_norm.argtypes = [vp, vp, i64, i64, vp, vp, vp]
    lib.r6_minibatch_adv_norm.restype = C.c_int
    lib.r6_adam_clip_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, vp, f32, f32, f32, vp, i64, i64, f32, vp]
    lib.r6_adam_clip_step.restype = C.c_int
    lib.r6_gae.argtypes = [vp, vp, vp, vp, i64, i64, f32, f32, vp, C.c_int, vp, vp, vp, vp]
    lib.r6_gae.restype = C.c_int
    lib.r6_permutation.argtypes = [vp, i64, u64, u64, vp, vp]
    lib.r6_permutation.restype = C.c_int
    lib.r6_tune.argtypes = [C.c_char_p, C.c_double]
    lib.r6_tune.restype = C.c_int
    lib.r6_rollout_size.restype = C.c_size_t
    lib.r6_rollout.argtypes = [C.POINTER(R6Params), C.POINTER(R6Buffers), C.POINTER(R6PolicyNet), C.POINTER(R6Rollout), u64, u64, u64, u64,
                               vp, u64, vp]
    lib.r6_rollout.restype = C.c_int
    lib.r6_device_status.restype = C.c_int
    lib.r6_clear_device_error.restype = C.c_int
    if lib.r6_rollout_size() != C.sizeof(R6Rollout):
        raise ImportError("R6Rollout layout mismatch between params.py and libr6dof.so: rebuild")
    if lib.r6_policy_grad_size() != C.sizeof(R6PolicyGrad):
        raise ImportError("R6PolicyGrad layout mismatch between params.py and libr6dof.so: rebuild")
    for name in ("r6_step", "r6_step_k", "r6_reset", "r6_observe", "r6_sim_step", "r6_rhs", "r6_policy_act", "r6_ppo_loss_grad"):
        getattr(lib, name).restype = C.c_int
    if lib.r6_abi_version() != ABI_VERSION:
        raise ImportError(f"libr6dof.so ABI {lib.r6_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    if lib.r6_params_size() != C.sizeof(R6Params) or lib.r6_buffers_size() != C.sizeof(R6Buffers):
        raise ImportError("R6Params/R6Buffers layout mismatch between params.py and libr6dof.so: rebuild")
    if lib.r6_stepk_size() != C.sizeof(R6StepK):
        raise ImportError("R6StepK layout mismatch between params.py and libr6dof.so: rebuild")
    if lib.r6_policy_net_size() != C.sizeof(R6PolicyNet):
        raise ImportError("R6PolicyNet layout mismatch between params.py and libr6dof.so: rebuild")
    _LIBS[path] = lib
    return lib


def check(rc: int, what: str = "r6", lib=None):
    """Raises R6Error for a non-zero return code.  ``lib`` = the handle the failing call went through (the error
    text is thread-local state of THAT library instance; A/B builds loaded side by side each have their own)."""
    if rc != 0:
        msg = (lib if lib is not None else load()).r6_last_error().decode(errors="replace")
        raise R6Error(f"{what} failed (rc={rc}): {msg}")


def tune(key: str, value: float, lib=None):
    """Sets a launch-shape knob of ``r6_step`` (include/r6dof.h ``r6_tune``); results never depend on it."""
    lib = lib if lib is not None else load()
    check(lib.r6_tune(key.encode(), float(value)), "r6_tune", lib)


def device_status(lib=None) -> int:
    """0, or R6_E_DEVICE while the library's device error word is raised (include/r6dof.h ``r6_device_status``)."""
    return int((lib if lib is not None else load()).r6_device_status())


def clear_device_error(lib=None) -> int:
    """Clears the device error word (after the caller has re-initialised its envs); returns the status it had."""
    return int((lib if lib is not None else load()).r6_clear_device_error())
