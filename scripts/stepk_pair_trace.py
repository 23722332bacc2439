#!/usr/bin/env python3
"""Per-wave timeline of r6::step_k_pair_kernel (dev tool, GPU box; -DR6_TRACE_K build): entry, loop start / end and end of every
wave, episode ends seen by the dynamics waves, shader cycles spent waiting for the partner."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, nargs="+", default=[16, 64])
ap.add_argument("--lib", default="rl_rocket_6dof_amd/csrc/variants/libr6dof_tracek.so")
a = ap.parse_args()
dev = torch.device("cuda:0"); n = 65536
env = BatchedRocket6DOF(n, default_env_config(), device=dev, integrator="euler", timestep=0.04, record_flags=False, track_episodes=True, seed=42, lib_path=a.lib)
env.reset()
gen = torch.Generator(device=dev).manual_seed(1)
acts = torch.rand((max(a.k), n, 3), generator=gen, device=dev) * 2 - 1
for _ in range(30):
    env.step_k(acts, last_only=True)
for K in a.k:
    out = env.step_k(acts[:K])
    for rep in range(3):
        env.stats.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_k(acts[:K], out=out); e1.record(); torch.cuda.synchronize()
        st = env.stats.cpu().numpy()[: (n // 64 // 4) * 16]
        st = st[(st[:, 0] != 0) | (st[:, 3] != 0)]
        base = st[:, 0].min(); us = lambda x: (x - base) / 100.0
        for role, name in ((0, "A dynamics"), (1, "B reward  "), (2, "C outputs ")):
            m = st[:, 6] == role
            t0, t1, t2, t3 = us(st[m, 0]), us(st[m, 1]), us(st[m, 2]), us(st[m, 3])
            print(f"K={K:3d} rep {rep} {name}: launch {e0.elapsed_time(e1) * 1e3:6.1f} us | entry {t0.min():.2f}..{t0.max():.2f} | loop start {t1.mean():.2f} (max {t1.max():.2f}) | "
                  f"loop {np.mean(t2 - t1):.2f} us mean, {np.max(t2 - t1):.2f} max = {np.mean(t2 - t1) / K:.3f} us/step | loop end max {t2.max():.2f} | wave end max {t3.max():.2f} | "
                  f"waiting for the partner {st[m, 5].mean() / K:.0f} cycles/step (max wave {st[m, 5].max() / K:.0f})" + (f" | episode ends/wave {st[m, 4].mean():.2f} max {st[m, 4].max():.0f}, corr(loop, ends) {np.corrcoef(t2 - t1, st[m, 4])[0, 1]:.2f}" if role == 0 else ""))
