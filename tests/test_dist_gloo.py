"""World-size-2 gloo tests of the multi-GPU layer (env sharding + the one all-gather).  CPU-only: the
collective and the shard arithmetic are exercised for real; the per-shard stepping is stood in for by
the C oracle (same RNG stream definition as the kernel), which lets the sharding-invariance property
be checked end to end: 2 shards with env_offset == 1 big batch, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import oracle_c as oc
    from rl_rocket_6dof_amd import default_env_config
    from rl_rocket_6dof_amd import dist as r6dist
    from rl_rocket_6dof_amd.params import STAT_NAMES

    r, w, _ = r6dist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    off, cnt = r6dist.shard_range(total, rank, world)
    env = oc.BatchOracle(cnt, default_env_config(), seed=7, env_offset=off, auto_reset=True, track=True)
    env.reset(step_index=1 << 60)
    rng = np.random.RandomState(123)
    acts = rng.uniform(-1, 1, (140, total, 3)).astype(np.float32)     # same global action table on every rank
    stats = dict.fromkeys(STAT_NAMES, 0.0)
    for k in range(140):
        pre_ret, pre_len = env.ep_return.copy(), env.ep_step.copy()
        env.step(acts[k, off:off + cnt])
        d = env.done.astype(bool)
        stats["episodes"] += d.sum()
        stats["ground"] += ((env.flags & 1) != 0)[d].sum()
        stats["bounds"] += ((env.flags & 2) != 0)[d].sum()
        stats["return_sum"] += (pre_ret + env.reward)[d].sum()
        stats["length_sum"] += (pre_len + 1)[d].sum()
    tot, table = r6dist.gather_episode_stats(stats)
    assert table.shape == (world, len(STAT_NAMES))
    # the tensor form used inside rollout loops (device tensors under RCCL; plain CPU tensors here)
    dev_table = r6dist.gather_episode_stats_device(torch.tensor([stats[k] for k in STAT_NAMES], dtype=torch.float64))
    assert dev_table.shape == (world, len(STAT_NAMES)) and torch.equal(dev_table, table)
    assert abs(table[rank, 0].item() - stats["episodes"]) < 0.5
    q.put((rank, off, cnt, env.state.copy(), tot, r6dist.summarize(tot)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_allgather():
    total, world = 301, 2           # odd on purpose: ragged shards (151 + 150)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, off0, cnt0, st0, tot0, sum0), (r1, off1, cnt1, st1, tot1, sum1) = res
    assert (off0, cnt0, off1, cnt1) == (0, 151, 151, 150)
    assert tot0 == tot1 and tot0["episodes"] > 0        # every rank holds the same global totals
    assert abs(sum0["ep_len_mean"] - tot0["length_sum"] / tot0["episodes"]) < 1e-9

    # sharding invariance: one unsharded batch visits exactly the same states
    from oracle import oracle_c as oc
    from rl_rocket_6dof_amd import default_env_config

    env = oc.BatchOracle(total, default_env_config(), seed=7, env_offset=0, auto_reset=True, track=True)
    env.reset(step_index=1 << 60)
    acts = np.random.RandomState(123).uniform(-1, 1, (140, total, 3)).astype(np.float32)
    for k in range(140):
        env.step(acts[k])
    np.testing.assert_array_equal(env.state, np.concatenate([st0, st1]))


def test_shard_range_properties():
    from rl_rocket_6dof_amd.dist import shard_range

    for total in (1, 7, 8, 1 << 20, (1 << 23) + 5):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_gather_without_process_group_is_identity():
    from rl_rocket_6dof_amd import dist as r6dist
    from rl_rocket_6dof_amd.params import STAT_NAMES

    local = {k: float(i + 1) for i, k in enumerate(STAT_NAMES)}
    tot, table = r6dist.gather_episode_stats(local)
    assert tot == local and table.shape == (1, len(STAT_NAMES))
