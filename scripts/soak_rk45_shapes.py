#!/usr/bin/env python3
"""Soak of the adaptive step kernel over launch shapes (dev tool, GPU box): the dispatcher-balanced grid at 2^22 envs (four tiles per wave), a
pinned small grid (many tiles per wave: the guest list is recycled all the time), dt = 0.5 (guests under way for several tiles); state
invariants checked along the way."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config, _lib
for n, grid, dt, steps in ((1 << 22, 0, 0.1, 1500), (300_011, 24, 0.1, 4000), (200_003, 32, 0.5, 2000)):
    _lib.tune("grid_blocks", grid)
    env = BatchedRocket6DOF(n, default_env_config(), integrator="rk45", timestep=dt, seed=9, track_episodes=True)
    env.reset()
    g = torch.Generator(device=env.device).manual_seed(5)
    pool = [torch.rand((n, 3), generator=g, device=env.device) * 2 - 1 for _ in range(3)]
    t0 = time.time()
    for k in range(steps):
        env.step_async(pool[k % 3])
        if (k + 1) % (steps // 4) == 0:
            st = env.get_state()
            assert bool(torch.isfinite(st).all()), "non-finite state"
            assert float((st[:, 6:10].norm(dim=1) - 1).abs().max()) < 1e-5
            assert float(st[:, 0].min()) > -31.0 and float(st[:, 13].min()) > 0.0
            s = env.episode_stats()
            assert s["ground"] + s["bounds"] >= s["episodes"] > 0
            print(f"n={n} grid={grid} dt={dt} step {k + 1}: ok, {s['episodes']:.0f} episodes, {time.time() - t0:.1f} s", flush=True)
    del env, pool
_lib.tune("grid_blocks", 0)
print("ok")
