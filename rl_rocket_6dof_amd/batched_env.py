"""``BatchedRocket6DOF`` -- N independent Rocket6DOF environments stepped by one HIP kernel launch.

Surface: stable-baselines3 ``VecEnv`` (SB3 1.6.0 public API: ``num_envs``, ``observation_space``,
``action_space``, ``reset``, ``step_async`` / ``step_wait`` / ``step``, ``close``, ``seed``,
``get_attr`` / ``set_attr`` / ``env_method`` / ``env_is_wrapped``), with the semantics the reference
relies on implicitly through ``DummyVecEnv`` + ``Monitor`` + ``TimeLimit`` (main_6DOF.py:44-53):
auto-reset on done with ``terminal_observation``, ``TimeLimit.truncated`` and ``episode`` {r,l}
in ``infos``.  Per-env semantics are those of ``Rocket6DOF`` (reference
my_environment/envs/rocket_env.py:180-231).

All arrays live in HBM as torch tensors (state as structure-of-arrays ``[14][ld]``) and are handed
to the kernel by pointer; ``step`` returns device tensors unless ``numpy_io=True`` (SB3 mode).
There is no CPU path: without the HIP library / a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Any, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .compat import Box, VecEnv
from .params import (ACTION_DIM, K_LAST_ONLY, MAX_BLOCKS, FLAG_BOUNDS, FLAG_GROUND, FLAG_TRUNCATED, NUM_STATS, NUM_TERMS,
                     STATE_DIM, STAT_NAMES, R6Buffers, R6Rollout, R6StepK, default_env_config, make_params)

TERM_NAMES_ACC = ("atarg_tracking", "thrust_penalty", "eta", "attitude_constraint", "goal_conditions",
                  "final_position", "final_velocity",          # rocket_env.py:342-359
                  "vertical_attitude_reward")                   # wrappers.py:141-147 (0 unless enabled)
TERM_NAMES_VEL = ("vtarg_tracking",) + TERM_NAMES_ACC[1:]

_EXPLICIT_RESET_BASE = 1 << 60   # keeps reset() draws disjoint from in-kernel auto-reset draws


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _leading_dim(n: int) -> int:
    """Row pitch (in floats) of the SoA arrays.  Rows a multiple of 8 KiB apart put the 14 state rows of a
    64-env tile on the same HBM channels; such pitches are skewed by 4 KiB (A/B at N = 2^20, pitch 4 MiB:
    40.6 -> 39.3 us per step).  ``R6_LD_PAD`` (floats) overrides the skew for experiments."""
    ld = (n + 63) // 64 * 64
    pad = os.environ.get("R6_LD_PAD")
    if pad is not None:
        return ld + int(pad)
    return ld + 1024 if ld % 2048 == 0 else ld


def group_bounds(n: int, groups: int) -> List[int]:
    """Env group g of ``BatchedRocket6DOF(groups=G)`` = envs [bounds[g], bounds[g + 1]): near-equal contiguous ranges cut at multiples
    of 64 (tile-aligned SoA rows; 16-byte aligned action / observation blocks whatever the observation width)."""
    if groups < 1 or (groups > 1 and n < 128 * groups):
        raise ValueError("groups must be >= 1 and every group needs at least two 64-env tiles")
    return [min(n, (n * g // groups + 32) // 64 * 64) for g in range(groups)] + [n]


class StaleInfosError(RuntimeError):
    pass


class LazyInfos(Sequence):
    """``infos`` of one step: behaves like SB3's list of N dicts, but dicts are built on access and
    the device data is copied to the host once, on first access (N = 2^20 dicts per step would
    dominate everything otherwise).

    The snapshot is taken from the env's live output buffers, which the NEXT step overwrites: an ``infos`` object
    belongs to the step that returned it.  Indexing it for the first time after the env has stepped again raises
    ``StaleInfosError`` instead of silently reporting the later step; call ``materialize()`` right after ``step``
    to keep an ``infos`` beyond its step (one device->host copy)."""

    def __init__(self, env: "BatchedRocket6DOF"):
        self._env = env
        self._host = None
        self._step = env._step_count

    def __len__(self):
        return self._env.num_envs

    @classmethod
    def from_host(cls, env: "BatchedRocket6DOF", views: dict) -> "LazyInfos":
        """The ``infos`` of the step whose outputs ``env.fetch_host()`` just brought over (copies of the pinned image: the
        next fetch overwrites it) -- no further device access."""
        self = cls(env)
        h = {"done": views["done"].astype(bool)}
        h["flags"] = views["flags"].copy() if "flags" in views else None
        idx = np.nonzero(h["done"])[0]
        h["done_idx"] = {int(k): j for j, k in enumerate(idx)}
        h["final_obs"] = h["fin_return"] = h["fin_length"] = None
        if len(idx) and env.auto_reset:
            h["final_obs"] = views["final_obs"][idx]
            if "fin_return" in views and "fin_length" in views:
                h["fin_return"], h["fin_length"] = views["fin_return"][idx], views["fin_length"][idx]
        h["terms"] = views["terms"][:, : env.num_envs].copy() if "terms" in views else None
        h["t"] = round(time.time() - env._t_start, 6)
        self._host = h
        return self

    def materialize(self) -> "LazyInfos":
        self._snapshot()
        return self

    def _snapshot(self):
        if self._host is None:
            e = self._env
            if e._step_count != self._step:
                raise StaleInfosError(
                    f"infos of step {self._step} first read after step {e._step_count}: the env's output buffers "
                    "now hold the later step; index infos before stepping again or call infos.materialize()")
            h = {"done": e.done.cpu().numpy().astype(bool)}
            h["flags"] = e.flags.cpu().numpy() if e.flags is not None else None
            idx = np.nonzero(h["done"])[0]
            h["done_idx"] = {int(k): j for j, k in enumerate(idx)}
            h["final_obs"] = h["fin_return"] = h["fin_length"] = None
            if len(idx) and e.auto_reset:
                sel = torch.as_tensor(idx, device=e.device)
                h["final_obs"] = e.final_obs[sel].cpu().numpy()
                if e.fin_return is not None and e.fin_length is not None:   # None when episode tracking is off
                    h["fin_return"] = e.fin_return[sel].cpu().numpy()
                    h["fin_length"] = e.fin_length[sel].cpu().numpy()
            h["terms"] = e.terms[:, : e.num_envs].cpu().numpy() if e.terms is not None else None
            h["t"] = round(time.time() - e._t_start, 6)
            self._host = h
        return self._host

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        h = self._snapshot()
        e = self._env
        info = {"is_done": bool(h["done"][i])}                         # rocket_env.py:219
        if h["flags"] is not None:
            f = int(h["flags"][i])
            info["bounds_violation"] = bool(f & FLAG_BOUNDS)           # rocket_env.py:226
            info["ground_contact"] = bool(f & FLAG_GROUND)
            if e.max_episode_steps:
                info["TimeLimit.truncated"] = bool(f & FLAG_TRUNCATED)  # gym TimeLimit
        if h["terms"] is not None:
            rd = {k: float(h["terms"][t, i]) for t, k in enumerate(e.term_names)}
            info["rewards_dict"] = rd                                  # rocket_env.py:218
            info.update(rd)                                            # rocket_env.py:223
        if h["done"][i] and e.auto_reset:
            j = h["done_idx"][i]
            info["terminal_observation"] = h["final_obs"][j]           # SB3 DummyVecEnv
            if h["fin_return"] is not None:                            # gym Monitor (main_6DOF.py:51): r, l, t
                info["episode"] = {"r": float(h["fin_return"][j]), "l": int(h["fin_length"][j]), "t": h["t"]}
        return info


class BatchedRocket6DOF(VecEnv):
    """N Rocket6DOF environments on one MI355X; see module docstring.

    Derives from ``stable_baselines3.common.vec_env.VecEnv`` whenever SB3 is importable (``compat.VecEnv``), so
    SB3's ``isinstance(env, VecEnv)`` check takes it unwrapped; without SB3 the base is a stand-in with the same
    constructor contract and helpers."""

    metadata = {"render.modes": []}

    def __init__(self, num_envs: int, env_config: Optional[dict] = None, *, device=None,
                 integrator: str = "rk4", n_substeps: Optional[int] = None, timestep: Optional[float] = None,
                 remove_mass: bool = False, clip_reward=None, max_episode_steps: int = 0,
                 auto_reset: bool = True, seed: Optional[int] = None, env_offset: int = 0,
                 record_terms: bool = False, record_flags: bool = True, track_episodes: bool = True,
                 numpy_io: bool = False, lib_path: Optional[str] = None, reward_annealing: bool = False,
                 vertical_attitude_reward=None, record_nfev: bool = False, groups: int = 1, group_sync: str = "caller",
                 host_mirror: Optional[bool] = None):
        """``host_mirror`` (batches read back every step: the single-env drop-in, SB3's numpy stepping): every per-env buffer is a view
        of ONE device allocation with a pinned host copy of the same layout; ``fetch_host()`` brings a whole step's outputs over in
        one copy and one synchronisation instead of one round trip per tensor.  The kernels see the same buffers either way.
        ``None`` (default): on for ``numpy_io=True`` up to 16 384 envs (measured: calls per second x2 at 8 envs, x3.7 at 256, x2.5 at
        4096; at 65 536 the whole-buffer download loses to the per-tensor one), off otherwise.

        ``groups`` > 1 (round 5): ``step_async`` steps the batch as that many contiguous env groups, each with its own ``r6_step``
        launch on a stream the env owns -- launches of ONE stream serialise, launches of two streams overlap, and one group's
        ramp (first loads) and tail (last tiles, reset drain) then run under the other group's bulk.  Results do not depend on
        it: envs are independent and the reset RNG is keyed by the global env index (the sharding invariance of section 5 of
        DESIGN.md, one level down).  ``group_sync``: "caller" -- every ``step_async`` makes the group streams wait for the
        caller's current stream (the actions are ready) and ``step_wait`` makes the caller's stream wait for the groups: safe
        anywhere, two event records and four stream waits per step; "none" -- no ordering against the caller's stream at all:
        the groups are independent pipelines (what an asynchronous vector env with one policy call per group does);
        ``join_groups()`` orders the caller's stream behind them on request.  reset / step_k / rollout / set_state always join."""
        if num_envs < 1:
            raise ValueError("num_envs must be >= 1")
        if group_sync not in ("caller", "none"):
            raise ValueError("group_sync must be 'caller' or 'none'")
        self.groups, self.group_sync = int(groups), group_sync
        group_bounds(int(num_envs), self.groups)      # raises on a bad split before anything is allocated
        self.lib = _lib.load(lib_path)   # raises if the HIP extension is missing -- no fallback
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedRocket6DOF needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU implementation in this package")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise ValueError("device must be a cuda (ROCm) device")
        self.env_config = dict(env_config) if env_config is not None else default_env_config()
        self.params, self.spec = make_params(
            self.env_config, integrator=integrator, n_substeps=n_substeps, timestep=timestep,
            remove_mass=remove_mass, clip_reward=clip_reward, max_episode_steps=max_episode_steps,
            auto_reset=auto_reset, reward_annealing=reward_annealing, vertical_attitude_reward=vertical_attitude_reward)
        self.obs_dim = int(self.params.obs_dim)
        self.auto_reset = bool(auto_reset)
        self.max_episode_steps = int(max_episode_steps)
        self.numpy_io = bool(numpy_io)
        self.env_offset = int(env_offset)
        self._seed = int(self.spec.seed if seed is None else seed)
        self.term_names = TERM_NAMES_ACC if self.spec.shaping_type == "acceleration" else TERM_NAMES_VEL

        VecEnv.__init__(self, int(num_envs),
                        Box(low=-1, high=1, shape=(self.obs_dim,), dtype=np.float32),   # rocket_env.py:137, wrappers.py:123
                        Box(low=-1, high=1, shape=(ACTION_DIM,), dtype=np.float32))     # rocket_env.py:147
        self.remove_mass, self.clip_reward = bool(remove_mass), clip_reward
        self._t_start = time.time()
        self.reward_coefficients = self.spec.reward_coefficients
        self.state_normalizer = self.spec.state_normalizer
        self.timestep = float(self.params.dt)

        n, dev = self.num_envs, self.device
        self.ld = ld = _leading_dim(n)
        f32 = dict(dtype=torch.float32, device=dev)
        self.state = torch.zeros((STATE_DIM, ld), **f32)
        self.epconst = torch.zeros((2, ld), **f32)
        self._act = torch.zeros((n, ACTION_DIM), **f32)
        self.obs = torch.zeros((n, self.obs_dim), **f32)
        self.reward = torch.zeros((n,), **f32)
        self.done = torch.zeros((n,), dtype=torch.uint8, device=dev)
        self.flags = torch.zeros((n,), dtype=torch.uint8, device=dev) if record_flags else None
        self.terms = torch.zeros((NUM_TERMS, ld), **f32) if record_terms else None
        self.nfev = torch.zeros((n,), dtype=torch.uint8, device=dev) if record_nfev else None
        self.final_obs = torch.zeros((n, self.obs_dim), **f32) if auto_reset else None
        track = bool(track_episodes or max_episode_steps)
        self.ep_step = torch.zeros((n,), dtype=torch.int32, device=dev) if track else None
        self.ep_return = torch.zeros((n,), **f32) if track else None
        self.fin_return = torch.zeros((n,), **f32) if (track and auto_reset) else None
        self.fin_length = torch.zeros((n,), dtype=torch.int32, device=dev) if (track and auto_reset) else None
        # float64: counts stay exact.  One block of MAX_BLOCKS per-workgroup rows per env group: concurrent launches never share a row
        self.stats = torch.zeros((MAX_BLOCKS * self.groups, NUM_STATS), dtype=torch.float64, device=dev) if track else None
        # group g = envs [bounds[g], bounds[g + 1]): cut at multiples of 64 (tile-aligned rows, 16-byte aligned action / obs blocks)
        self._group_bounds = group_bounds(n, self.groups)
        self._group_streams = [torch.cuda.Stream(device=dev) for _ in range(self.groups)] if self.groups > 1 else []
        self._fork_event = torch.cuda.Event() if self.groups > 1 else None
        self._join_events = [torch.cuda.Event() for _ in range(self.groups)] if self.groups > 1 else []
        self._groups_in_flight = False

        self._arena = self._host_arena = None
        self._host_views: dict = {}
        # numpy_io (the SB3 way of stepping: numpy in, numpy out, every step): a batch small enough to be read back whole takes the
        # mirror by itself -- step_wait is then one download of the output buffers instead of a round trip per tensor
        if host_mirror or (host_mirror is None and self.numpy_io and n <= (1 << 14)):
            self._build_host_mirror()
        self._buf = R6Buffers()
        self._gbufs = []
        self._step_base = None
        self._fill_buffers(self._act)
        self._step_count = 0     # global step counter (keys the auto-reset RNG), first step = 1
        self._reset_count = 0
        self._pending = None
        self._closed = False

    # ------------------------------------------------------------------------------------------
    def _fill_buffers(self, action: torch.Tensor):
        b = self._buf
        b.n, b.ld = self.num_envs, self.ld
        b.state, b.epconst = _ptr(self.state), _ptr(self.epconst)
        b.action, b.obs, b.reward, b.done = _ptr(action), _ptr(self.obs), _ptr(self.reward), _ptr(self.done)
        b.flags, b.terms, b.final_obs = _ptr(self.flags), _ptr(self.terms), _ptr(self.final_obs)
        b.ep_step, b.ep_return = _ptr(self.ep_step), _ptr(self.ep_return)
        b.fin_return, b.fin_length, b.stats = _ptr(self.fin_return), _ptr(self.fin_length), _ptr(self.stats)
        b.nfev = _ptr(self.nfev)
        if self.groups > 1 and hasattr(self, "_group_bounds"):
            self._gbufs = [self._group_buffers(b, g) for g in range(self.groups)]   # per-step work: only their action pointers move

    # outputs of a step first (what step_wait hands out: one contiguous range), then the env's own state
    _MIRRORED_OUT = ("obs", "reward", "done", "flags", "terms", "nfev", "final_obs", "fin_return", "fin_length")
    _MIRRORED = _MIRRORED_OUT + ("state", "epconst", "_act", "ep_step", "ep_return")

    def _build_host_mirror(self) -> None:
        """Moves the (still zero) per-env buffers into one device allocation, 256-byte aligned each, next to a pinned host
        image of the same layout with numpy views of every buffer (``_host_views``)."""
        plan, off = [], 0
        for name in self._MIRRORED:
            t = getattr(self, name)
            if t is not None:
                nbytes = t.numel() * t.element_size()
                plan.append((name, off, nbytes, t.dtype, tuple(t.shape)))
                off += (nbytes + 255) // 256 * 256
                if name in self._MIRRORED_OUT:
                    self._out_bytes = off
        self._arena = torch.zeros((off,), dtype=torch.uint8, device=self.device)
        self._host_arena = torch.zeros((off,), dtype=torch.uint8).pin_memory()
        self._host_action = torch.zeros((self.num_envs, ACTION_DIM), dtype=torch.float32).pin_memory()
        self._host_action_np = self._host_action.numpy()
        self._act_event = torch.cuda.Event()
        for name, o, nbytes, dtype, shape in plan:
            setattr(self, name, self._arena[o:o + nbytes].view(dtype).view(shape))
            self._host_views[name] = self._host_arena[o:o + nbytes].view(dtype).view(shape).numpy()

    def fetch_host(self, outputs_only: bool = False) -> dict:
        """``host_mirror=True``: one device->host copy of every per-env buffer (``outputs_only``: of a step's outputs -- obs, reward,
        done, flags, terms, nfev, final_obs, fin_return, fin_length; the other views keep what the last full fetch brought)
        behind whatever the current stream holds, one synchronisation; returns numpy views of the pinned image, keyed by
        buffer name (valid until the next call)."""
        if self._arena is None:
            raise RuntimeError("fetch_host() needs an env built with host_mirror=True")
        self.join_groups()
        if outputs_only:
            self._host_arena[: self._out_bytes].copy_(self._arena[: self._out_bytes], non_blocking=True)
        else:
            self._host_arena.copy_(self._arena, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self._host_views

    def step_host(self, actions) -> dict:
        """``host_mirror=True``: one step driven from the host -- the (N, 3) actions go up through a pinned buffer, every output
        comes back through ``fetch_host()``: one launch, two small copies and one synchronisation per step."""
        if self._arena is None:
            raise RuntimeError("step_host() needs an env built with host_mirror=True")
        self.step_async(np.asarray(actions, dtype=np.float32))
        self._pending = None
        return self.fetch_host()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _out(self, t: torch.Tensor):
        return t.cpu().numpy() if self.numpy_io else t

    # ------------------------------------------------------------------------------------------
    def seed(self, seed: Optional[int] = None):
        """rocket_env.py:619-621 -- reseeds the initial-condition stream."""
        if seed is not None:
            self._seed = int(seed)
            self._reset_count = 0
        return [self._seed + self.env_offset + i for i in range(min(self.num_envs, 8))]

    def reset(self, mask: Optional[torch.Tensor] = None):
        """rocket_env.py:180-199 for every env (or those with mask != 0): sample the initial
        condition from U(IC - ICRange/2, IC + ICRange/2), normalise the quaternion, new episode."""
        self.join_groups()
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            if m.shape != (self.num_envs,):
                raise ValueError("mask must have shape (num_envs,)")
        with torch.cuda.device(self.device):
            rc = self.lib.r6_reset(C.byref(self.params), C.byref(self._buf), _ptr(m), self._seed,
                                   _EXPLICIT_RESET_BASE + self._reset_count, self.env_offset, self._stream())
        _lib.check(rc, "r6_reset", self.lib)
        self._reset_count += 1
        if m is not None:
            torch.cuda.current_stream(self.device).synchronize()  # keep `m` alive until consumed
        return self._out(self.obs)

    def _group_buffers(self, b: R6Buffers, g: int) -> R6Buffers:
        """The R6Buffers of env group g: every pointer moved to the group's first env (SoA rows keep their pitch ``ld``), its
        own block of statistics rows."""
        a0, a1 = self._group_bounds[g], self._group_bounds[g + 1]
        od = self.obs_dim
        gb = R6Buffers.from_buffer_copy(b)
        gb.n = a1 - a0

        def mv(name, bytes_per_env):
            v = getattr(b, name)
            if v:
                setattr(gb, name, C.c_void_p(v + a0 * bytes_per_env))

        for name, w in (("state", 4), ("epconst", 4), ("action", 4 * ACTION_DIM), ("obs", 4 * od), ("reward", 4), ("done", 1), ("flags", 1),
                        ("terms", 4), ("final_obs", 4 * od), ("ep_step", 4), ("ep_return", 4), ("fin_return", 4), ("fin_length", 4), ("nfev", 1)):
            mv(name, w)
        if b.stats:
            gb.stats = C.c_void_p(b.stats + g * MAX_BLOCKS * NUM_STATS * 8)
        return gb

    def join_groups(self) -> None:
        """groups > 1: the caller's current stream waits for everything the group streams have been given so far."""
        if self._groups_in_flight:
            cur = torch.cuda.current_stream(self.device)
            for ev, st in zip(self._join_events, self._group_streams):
                ev.record(st)
                cur.wait_event(ev)
            self._groups_in_flight = False

    def step_async(self, actions) -> None:
        if torch.is_tensor(actions) and actions.device == self.device and actions.dtype == torch.float32 \
                and actions.is_contiguous() and actions.data_ptr() % 16 == 0:
            a = actions          # zero-copy
        elif self._arena is not None and not torch.is_tensor(actions):     # host actions through the pinned buffer: no staging copy
            self._act_event.synchronize()      # the previous upload has left the pinned buffer (a no-op when the caller synchronised)
            self._host_action_np[...] = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, ACTION_DIM)
            self._act.copy_(self._host_action, non_blocking=True)
            self._act_event.record(torch.cuda.current_stream(self.device))
            a = self._act
        else:
            self._act.copy_(torch.as_tensor(actions, dtype=torch.float32).reshape(self.num_envs, ACTION_DIM))
            a = self._act
        if tuple(a.shape) != (self.num_envs, ACTION_DIM):
            raise ValueError(f"actions must have shape ({self.num_envs}, {ACTION_DIM})")
        self._buf.action = _ptr(a)
        self._step_count += 1
        with torch.cuda.device(self.device):
            if self.groups == 1:
                rc = self.lib.r6_step(C.byref(self.params), C.byref(self._buf), self._seed, self._step_count,
                                      self.env_offset, self._stream())
                _lib.check(rc, "r6_step", self.lib)
            else:
                if self.group_sync == "caller":     # fork: the groups start behind whatever the caller's stream holds (the actions)
                    self._fork_event.record(torch.cuda.current_stream(self.device))
                for g, st in enumerate(self._group_streams):
                    if self.group_sync == "caller":
                        st.wait_event(self._fork_event)
                    gb = self._gbufs[g]
                    gb.action = a.data_ptr() + self._group_bounds[g] * 4 * ACTION_DIM
                    gb.step_base = self._buf.step_base
                    rc = self.lib.r6_step(C.byref(self.params), C.byref(gb), self._seed, self._step_count,
                                          self.env_offset + self._group_bounds[g], C.c_void_p(st.cuda_stream))
                    _lib.check(rc, "r6_step", self.lib)
                self._groups_in_flight = True
        self._pending = a   # keeps the action tensor alive until step_wait

    def set_step_base(self, base: Optional[torch.Tensor]) -> None:
        """``R6Buffers.step_base``: an int64 device scalar that the kernels ADD to the step counter keying the reset RNG.
        A step recorded into a HIP graph replays with the host counter it was captured with; the caller moves ``base`` on
        between replays (``base += steps per replay``, itself a node of the graph) and finished episodes keep drawing
        fresh initial conditions.  ``None`` detaches it."""
        if base is not None and (base.dtype != torch.int64 or base.device != self.device or base.numel() != 1):
            raise ValueError("step base must be an int64 scalar tensor on the env's device")
        self._step_base = base          # keeps the tensor alive
        self._buf.step_base = _ptr(base)

    def step_into(self, actions: torch.Tensor, *, obs: Optional[torch.Tensor] = None, reward: Optional[torch.Tensor] = None,
                  done: Optional[torch.Tensor] = None, flags: Optional[torch.Tensor] = None,
                  final_obs: Optional[torch.Tensor] = None) -> None:
        """One env step whose outputs go straight into the caller's tensors (a rollout buffer's rows for this time step)
        instead of the env's own ``obs`` / ``reward`` / ``done`` / ``flags`` / ``final_obs`` -- no copy kernels in a rollout
        loop, and nothing but the launch itself when the loop is recorded into a HIP graph.  Outputs that are not given
        go to the env's buffers as in ``step``.  ``done`` / ``flags``: uint8; every tensor contiguous on the env's device,
        ``obs`` / ``final_obs`` 16-byte aligned.  Device tensors in, nothing returned, no ``infos``."""
        self.join_groups()
        n, od = self.num_envs, self.obs_dim
        if not (torch.is_tensor(actions) and actions.device == self.device and actions.dtype == torch.float32
                and actions.is_contiguous() and tuple(actions.shape) == (n, ACTION_DIM) and actions.data_ptr() % 16 == 0):
            raise ValueError(f"step_into needs a contiguous, 16-byte aligned float32 ({n}, {ACTION_DIM}) action tensor on {self.device}")

        def chk(t, name, shape, dtype, align):
            if t is None:
                return None
            if t.device != self.device or t.dtype != dtype or tuple(t.shape) != shape or not t.is_contiguous() or t.data_ptr() % align:
                raise ValueError(f"step_into: {name} must be a contiguous {dtype} {shape} tensor on {self.device}, {align}-byte aligned")
            return t

        b = R6Buffers.from_buffer_copy(self._buf)
        b.action = _ptr(actions)
        if chk(obs, "obs", (n, od), torch.float32, 16) is not None:
            b.obs = _ptr(obs)
        if chk(reward, "reward", (n,), torch.float32, 4) is not None:
            b.reward = _ptr(reward)
        if chk(done, "done", (n,), torch.uint8, 1) is not None:
            b.done = _ptr(done)
        if chk(flags, "flags", (n,), torch.uint8, 1) is not None:
            if self.flags is None:
                raise ValueError("step_into: flags need an env built with record_flags=True")
            b.flags = _ptr(flags)
        if chk(final_obs, "final_obs", (n, od), torch.float32, 16) is not None:
            if self.final_obs is None:
                raise ValueError("step_into: final_obs needs an env that records terminal observations")
            b.final_obs = _ptr(final_obs)
        self._step_count += 1
        with torch.cuda.device(self.device):
            rc = self.lib.r6_step(C.byref(self.params), C.byref(b), self._seed, self._step_count, self.env_offset, self._stream())
        _lib.check(rc, "r6_step", self.lib)

    def step_wait(self):
        if self.groups > 1 and self.group_sync == "caller":
            self.join_groups()
        self._pending = None
        if self.numpy_io and self._arena is not None:
            h = self.fetch_host(outputs_only=True)
            return h["obs"].copy(), h["reward"].copy(), h["done"].astype(bool), LazyInfos.from_host(self, h)
        done = self.done.view(torch.bool)
        return self._out(self.obs), self._out(self.reward), self._out(done), LazyInfos(self)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def step_k(self, actions, *, last_only: bool = False, out: Optional[dict] = None) -> dict:
        """K env steps in ONE kernel launch (``r6_step_k``): the per-timestep loop of the reference
        (main_6DOF.py:136-139, montecarlo_script.py:33-51) for open-loop action sequences -- random-action rollouts,
        replayed trajectories, burn-in -- without a launch per step.  ``actions``: device tensor (K, N, 3).

        Returns a dict of device tensors with a leading K dimension: ``obs`` (K,N,obs_dim), ``reward`` (K,N), ``done``
        (K,N) and, when the env records them, ``flags``, ``terms`` (K,8,ld), ``nfev``, ``final_obs`` (K,N,obs_dim; rows of
        envs that finished in that step), ``fin_return`` / ``fin_length`` (K,N).  Results are bit for bit those of K
        ``step`` calls.  ``last_only=True`` keeps just the last step's outputs (in the env's own single-step buffers:
        ``env.obs`` ...) and stores nothing for the earlier ones.  ``out`` recycles a dict returned by an earlier call
        with the same K."""
        self.join_groups()
        a = actions
        if not (torch.is_tensor(a) and a.device == self.device and a.dtype == torch.float32 and a.is_contiguous()
                and a.dim() == 3 and tuple(a.shape[1:]) == (self.num_envs, ACTION_DIM) and a.data_ptr() % 16 == 0):
            a = torch.as_tensor(actions, dtype=torch.float32, device=self.device).reshape(-1, self.num_envs, ACTION_DIM).contiguous()
        K, n, od, dev = int(a.shape[0]), self.num_envs, self.obs_dim, self.device
        if K < 1:
            raise ValueError("step_k needs at least one step of actions")
        sk = R6StepK(k=K, flags=K_LAST_ONLY if last_only else 0, action_stride=n * ACTION_DIM)
        b = R6Buffers.from_buffer_copy(self._buf)
        b.action = _ptr(a)
        if last_only:
            res = {"obs": self.obs, "reward": self.reward, "done": self.done.view(torch.bool)}
        else:
            res = out if out is not None and out["reward"].shape[0] == K else None
            if res is None:
                f32 = dict(dtype=torch.float32, device=dev)
                u8 = dict(dtype=torch.uint8, device=dev)
                pitch = (n * od + 3) // 4 * 4     # per-step blocks start 16-byte aligned (the kernel stores 16 B per lane)
                rows = lambda: torch.zeros((K, pitch), **f32)[:, : n * od].view(K, n, od)
                res = {"obs": rows(), "reward": torch.empty((K, n), **f32), "_done_u8": torch.empty((K, n), **u8)}
                if self.flags is not None:
                    res["flags"] = torch.empty((K, n), **u8)
                if self.terms is not None:
                    res["terms"] = torch.zeros((K, NUM_TERMS, self.ld), **f32)
                if self.nfev is not None:
                    res["nfev"] = torch.empty((K, n), **u8)
                if self.final_obs is not None:
                    res["final_obs"] = rows()
                if self.fin_return is not None:
                    res["fin_return"] = torch.zeros((K, n), **f32)
                    res["fin_length"] = torch.zeros((K, n), dtype=torch.int32, device=dev)
                res["done"] = res["_done_u8"].view(torch.bool)
            b.obs, b.reward, b.done = _ptr(res["obs"]), _ptr(res["reward"]), _ptr(res["_done_u8"])
            b.flags, b.terms, b.nfev = _ptr(res.get("flags")), _ptr(res.get("terms")), _ptr(res.get("nfev"))
            b.final_obs, b.fin_return, b.fin_length = _ptr(res.get("final_obs")), _ptr(res.get("fin_return")), _ptr(res.get("fin_length"))
            sk.obs_stride, sk.final_obs_stride = res["obs"].stride(0), (res["final_obs"].stride(0) if "final_obs" in res else 0)
            sk.reward_stride = sk.done_stride = sk.flags_stride = sk.nfev_stride = sk.fin_stride = n
            sk.terms_stride = NUM_TERMS * self.ld
        with torch.cuda.device(dev):
            rc = self.lib.r6_step_k(C.byref(self.params), C.byref(b), C.byref(sk), self._seed, self._step_count + 1,
                                    self.env_offset, self._stream())
        _lib.check(rc, "r6_step_k", self.lib)
        self._step_count += K
        self._pending = a
        return res

    def rollout(self, policy, k: int, *, reward: torch.Tensor, done: torch.Tensor, obs_seq: Optional[torch.Tensor] = None,
                raw_action: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None, logp: Optional[torch.Tensor] = None,
                flags: Optional[torch.Tensor] = None, boot_value: Optional[torch.Tensor] = None, deterministic: bool = False,
                policy_step_index: int = 1, policy_step_base: Optional[torch.Tensor] = None) -> None:
        """``k`` closed-loop rollout steps in ONE launch (``r6_rollout``): per step the policy forward of ``policy`` (a
        ``policy.FusedActorCritic``), the clip to the action Box, this env's step, and V(terminal observation) for the envs a
        time limit truncated -- SB3's ``collect_rollouts`` loop behind main_6DOF.py:136-139.  Bit for bit the sequence
        ``policy.forward(obs, step_index=policy_step_index + j)`` -> ``step_into`` [-> masked forward] for j = 0..k-1.
        All outputs are contiguous device tensors with a leading ``k`` dimension: ``reward`` (k, N) float32, ``done`` (k, N) uint8,
        and optionally ``obs_seq`` (k + 1, N, obs_dim) (row j + 1 <- the observation after step j; row 0 is left alone),
        ``raw_action`` (k, N, 3), ``value`` / ``logp`` (k, N), ``flags`` (k, N) uint8, ``boot_value`` (k, N) (written only where a
        step truncated an env).  ``env.obs`` receives the observation after the last step."""
        self.join_groups()
        n, od, dev = self.num_envs, self.obs_dim, self.device
        if policy.obs_dim != od or policy.device != dev:
            raise ValueError("policy and env disagree on obs_dim / device")
        if getattr(policy, "env_offset", self.env_offset) != self.env_offset:
            # r6_rollout takes ONE env_offset for the reset stream and the action-noise stream; a policy built with another offset
            # would draw other noise in policy.forward() than in here, silently
            raise ValueError(f"rollout: policy.env_offset {policy.env_offset} != env.env_offset {self.env_offset}: the fused rollout keys "
                             "the action noise by the env's global index; build the policy with the env's offset")

        def chk(t, name, shape, dtype):
            if t is None:
                return None
            if t.device != dev or t.dtype != dtype or tuple(t.shape) != shape or not t.is_contiguous():
                raise ValueError(f"rollout: {name} must be a contiguous {dtype} {shape} tensor on {dev}")
            return C.c_void_p(t.data_ptr())

        ro = R6Rollout(k=int(k), deterministic=int(bool(deterministic)))
        ro.reward, ro.done = chk(reward, "reward", (k, n), torch.float32), chk(done, "done", (k, n), torch.uint8)
        ro.obs_seq = chk(obs_seq, "obs_seq", (k + 1, n, od), torch.float32)
        ro.raw_action = chk(raw_action, "raw_action", (k, n, 3), torch.float32)
        ro.value, ro.logp = chk(value, "value", (k, n), torch.float32), chk(logp, "logp", (k, n), torch.float32)
        ro.flags, ro.boot_value = chk(flags, "flags", (k, n), torch.uint8), chk(boot_value, "boot_value", (k, n), torch.float32)
        if policy_step_base is not None and (policy_step_base.dtype != torch.int64 or policy_step_base.device != dev or policy_step_base.numel() != 1):
            raise ValueError("policy_step_base must be an int64 scalar tensor on the env's device")
        net = policy._net()
        with torch.cuda.device(dev):
            rc = self.lib.r6_rollout(C.byref(self.params), C.byref(self._buf), C.byref(net), C.byref(ro), self._seed, self._step_count + 1,
                                     policy.seed, int(policy_step_index), None if policy_step_base is None else C.c_void_p(policy_step_base.data_ptr()),
                                     self.env_offset, self._stream())
        _lib.check(rc, "r6_rollout", self.lib)
        self._step_count += int(k)

    # ------------------------------------------------------------------------------------------
    def get_state(self) -> torch.Tensor:
        """(N,14) copy of the simulator states (rocket_env.py:623)."""
        self.join_groups()
        return self.state[:, : self.num_envs].t().contiguous()

    def set_state(self, states, m0=None, v0=None):
        """Loads explicit simulator states (N,14); m0 defaults to the state's mass (a fresh
        Simulator6DOF, simulator.py:42), v0 to its speed (rocket_env.py:649-651)."""
        self.join_groups()
        s = torch.as_tensor(states, dtype=torch.float32, device=self.device).reshape(self.num_envs, STATE_DIM)
        self.state[:, : self.num_envs] = s.t()
        self.epconst[0, : self.num_envs] = s[:, 13] if m0 is None else torch.as_tensor(m0, dtype=torch.float32, device=self.device)
        self.epconst[1, : self.num_envs] = s[:, 3:6].norm(dim=1) if v0 is None else torch.as_tensor(v0, dtype=torch.float32, device=self.device)
        if self.ep_step is not None:
            self.ep_step.zero_()
            self.ep_return.zero_()
        with torch.cuda.device(self.device):
            rc = self.lib.r6_observe(C.byref(self.params), C.byref(self._buf), self._stream())
        _lib.check(rc, "r6_observe", self.lib)
        return self._out(self.obs)

    def terminal_errors(self) -> dict:
        """Terminal-state errors of the envs that finished in the LAST step -- the quantities the
        reference's montecarlo_script.py:33-51 writes per episode (position / velocity / attitude /
        angular-velocity error and the terminal mass it labels "used mass").  Needs auto_reset (the
        terminal observation is kept in ``final_obs``); device tensors, one entry per finished env."""
        if self.final_obs is None:
            raise RuntimeError("terminal_errors() needs auto_reset=True")
        idx = self.done.nonzero(as_tuple=True)[0]
        norm = torch.as_tensor(self.state_normalizer[: self.obs_dim], dtype=torch.float32, device=self.device)
        st = self.final_obs[idx] * norm
        out = {
            "env_index": idx,
            "final_position_error": st[:, 0:3].norm(dim=1),
            "final_velocity_error": st[:, 3:6].norm(dim=1),
            "attitude_error": 0.5 * torch.rad2deg(torch.acos(st[:, 6].clamp(-1, 1))),
            "angular_velocity_error": st[:, 10:13].norm(dim=1),
        }
        if self.obs_dim == 14:
            out["used mass"] = st[:, 13]
        return out

    def episode_stats_tensor(self, clear: bool = True) -> torch.Tensor:
        """The sums of ``episode_stats`` as a float64 device tensor [NUM_STATS], without a host synchronisation
        (stream-ordered: one reduction kernel and, with ``clear``, one fill)."""
        if self.stats is None:
            raise RuntimeError("episode tracking is disabled (track_episodes=False)")
        self.join_groups()
        s = self.stats.sum(0)
        if clear:
            self.stats.zero_()
        return s

    def episode_stats(self, clear: bool = True) -> dict:
        """Sums over the episodes finished since the last clear (per-workgroup partials reduced in
        float64 on the host)."""
        if self.stats is None:
            raise RuntimeError("episode tracking is disabled (track_episodes=False)")
        self.join_groups()
        s = self.stats.sum(0).cpu().numpy()
        if clear:
            self.stats.zero_()
        return {k: float(s[i]) for i, k in enumerate(STAT_NAMES)}

    # ---- SB3 VecEnv plumbing ---------------------------------------------------------------------
    def close(self) -> None:
        self._closed = True

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        return [getattr(self, attr_name)] * len(self._indices(indices))

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        setattr(self, attr_name, value)

    def env_method(self, method_name: str, *args, indices=None, **kwargs) -> List[Any]:
        out = getattr(self, method_name)(*args, **kwargs)
        return [out] * len(self._indices(indices))

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        """True for the wrappers of the reference's make_env() stack (main_6DOF.py:44-53) whose fused equivalent is
        switched on -- e.g. SB3's evaluate_policy asks for ``Monitor`` and then reads ``info["episode"]``."""
        name = getattr(wrapper_class, "__name__", str(wrapper_class))
        fused = {"Monitor": self.fin_return is not None, "TimeLimit": self.max_episode_steps > 0,
                 "RemoveMassFromObs": self.remove_mass, "ClipReward": self.clip_reward is not None}
        return [bool(fused.get(name, False))] * len(self._indices(indices))

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def render(self, mode: str = "rgb_array"):
        raise NotImplementedError("rendering (pyvista) is outside the hot path and not provided")

    def __repr__(self):
        integ = {0: "euler", 1: "rk4", 2: "rk45"}[self.params.integrator]
        return (f"BatchedRocket6DOF(num_envs={self.num_envs}, {integ}, dt={self.timestep}, "
                f"substeps={self.params.n_substeps}, obs_dim={self.obs_dim}, device={self.device})")
