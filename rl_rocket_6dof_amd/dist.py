"""Multi-GPU: one process per GPU, environments sharded embarrassingly, no collective in ``step``.

The reference has no parallelism of any kind (n_envs = 1, main_6DOF.py:105-114); this is the build's
data-parallel layer.  Rank r of W owns the global envs [offset_r, offset_r + count_r); the reset RNG
is keyed by the GLOBAL env index, so a W-GPU run visits exactly the states of a 1-GPU run.  The one
collective -- an all-gather of a fixed-size per-rank episode summary, once per rollout -- goes through
``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` on CPU for tests).  The
message is NUM_STATS float64 per rank: latency-bound, nowhere near xGMI link bandwidth.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from .params import NUM_STATS, STAT_NAMES


def _collectives_on() -> bool:
    """True when a process group exists and has more than one rank -- or has one rank and R6_FORCE_PROCESS_GROUP=1 asks for
    the collectives all the same (a builder's rehearsal of the RCCL calls on a single GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("R6_FORCE_PROCESS_GROUP") == "1"


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """(offset, count) of rank's contiguous shard; counts differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("need 0 <= rank < world")
    q, r = divmod(total, world)
    count = q + (1 if rank < r else 0)
    offset = rank * q + min(rank, r)
    return offset, count


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialises torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world, local_rank); no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("R6_FORCE_DEVICE", local)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


_AGREE_GENERATION: Dict[str, int] = {}     # calls made so far per key: the ranks call in the same order, so their counters agree


def all_ranks_agree(ok: bool, key: str) -> bool:
    """True iff EVERY rank passed ok=True.  Goes through the default process group's key-value store (the TCP rendezvous
    store), not through a collective of any backend -- usable to decide which backend's collectives to use next.  Without a
    process group (or without access to its store): ``ok``.

    Every call writes keys of its own generation (``r6/<key>/<call number>/<rank>``): a second call with the same key never
    reads the answers of the first (``get`` on an existing key does not block, so stale entries would let ranks decide
    differently -- the split-backend hang this function exists to prevent)."""
    if not (dist.is_available() and dist.is_initialized()):
        return bool(ok)
    try:
        from torch.distributed import distributed_c10d as c10d

        store = c10d._get_default_store()       # private API: guarded
    except Exception:
        return bool(ok)
    rank, world = dist.get_rank(), dist.get_world_size()
    gen = _AGREE_GENERATION.get(key, 0)
    _AGREE_GENERATION[key] = gen + 1
    store.set(f"r6/{key}/{gen}/{rank}", "1" if ok else "0")
    return all(store.get(f"r6/{key}/{gen}/{r}") == b"1" for r in range(world))     # get() blocks until the key exists


def gather_episode_stats(local: Dict[str, float], device=None) -> Tuple[Dict[str, float], torch.Tensor]:
    """All-gathers the per-rank episode summaries (``BatchedRocket6DOF.episode_stats()``).
    Returns (global totals, per-rank table [world, NUM_STATS])."""
    vec = torch.tensor([local[k] for k in STAT_NAMES], dtype=torch.float64)
    if _collectives_on():
        if dist.get_backend() == "nccl":
            vec = vec.to(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        rows = [torch.empty_like(vec) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, vec)          # RCCL all-gather on GPU ranks, gloo on CPU ranks
        table = torch.stack(rows).cpu()
    else:
        table = vec.reshape(1, NUM_STATS)
    tot = table.sum(0)
    return {k: float(tot[i]) for i, k in enumerate(STAT_NAMES)}, table


def gather_episode_stats_device(vec: torch.Tensor) -> torch.Tensor:
    """All-gather of a rank's episode summary given as a float64 DEVICE tensor [NUM_STATS]
    (``BatchedRocket6DOF.episode_stats_tensor``): returns the per-rank table [world, NUM_STATS] on the same device.
    Under RCCL (backend nccl) nothing touches the host -- the collective is stream-ordered behind the kernels that
    produced the sums, so a rollout loop can issue it without draining the GPU; gloo goes through the host."""
    if not _collectives_on():
        return vec.reshape(1, -1)
    world = dist.get_world_size()
    if dist.get_backend() == "nccl":
        out = torch.empty((world, vec.numel()), dtype=vec.dtype, device=vec.device)
        dist.all_gather_into_tensor(out, vec.contiguous())
        return out
    rows = [torch.empty(vec.numel(), dtype=vec.dtype) for _ in range(world)]
    dist.all_gather(rows, vec.cpu())
    return torch.stack(rows).to(vec.device)


def reduce_bench_times(wall_s: float, kernel_ms: float, device=None, group=None) -> Dict[str, object]:
    """Min / max over ranks of a rank's wall time and per-step kernel time (bench.py: the job's time is the slowest
    rank's), plus the device index every rank ran on.  One all-gather of three float64 per rank: on the host through
    ``group`` when one is given (a gloo group: the numbers are host floats, and the job's record then does not depend on the
    device collectives coming up), else through the device for RCCL (backend nccl), on the host for gloo; the identity
    without a process group."""
    dev_index = torch.cuda.current_device() if torch.cuda.is_available() else -1
    vec = torch.tensor([wall_s, kernel_ms, float(dev_index)], dtype=torch.float64)
    if _collectives_on():
        if group is None and dist.get_backend() == "nccl":
            vec = vec.to(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        rows = [torch.empty_like(vec) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, vec, group=group)
        table = torch.stack(rows).cpu()
    else:
        table = vec.reshape(1, 3)
    return {"wall_max": float(table[:, 0].max()), "wall_min": float(table[:, 0].min()),
            "kernel_ms_max": float(table[:, 1].max()), "kernel_ms_min": float(table[:, 1].min()),
            "devices": [int(x) for x in table[:, 2].tolist()]}


def summarize(tot: Dict[str, float]) -> Dict[str, float]:
    n = max(tot["episodes"], 1.0)
    mean = tot["return_sum"] / n
    var = max(tot["return_sq_sum"] / n - mean * mean, 0.0)
    return {"episodes": tot["episodes"], "ep_rew_mean": mean, "ep_rew_std": var ** 0.5,
            "ep_len_mean": tot["length_sum"] / n, "ground_frac": tot["ground"] / n,
            "bounds_frac": tot["bounds"] / n, "truncated_frac": tot["truncated"] / n,
            "landed_frac": tot["landed"] / n}


def make_sharded_env(total_envs: int, env_config=None, **kwargs):
    """This rank's shard of a ``total_envs``-wide job as a ``BatchedRocket6DOF`` (env_offset set)."""
    from .batched_env import BatchedRocket6DOF

    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    offset, count = shard_range(total_envs, rank, world)
    return BatchedRocket6DOF(count, env_config, env_offset=offset, **kwargs)
