#!/usr/bin/env python3
"""Per-wave timeline of r6::step_k_kernel (dev tool, GPU box; needs the -DR6_TRACE_K build: scripts/build_variant.sh tracek -DR6_TRACE_K)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--k", type=int, nargs="+", default=[16, 64])
ap.add_argument("--lib", default="rl_rocket_6dof_amd/csrc/variants/libr6dof_tracek.so")
args = ap.parse_args()
dev = torch.device("cuda:0")
env = BatchedRocket6DOF(args.n, default_env_config(), device=dev, integrator="euler", timestep=0.04, record_flags=False,
                        track_episodes=True, seed=42, lib_path=args.lib)
env.reset()
gen = torch.Generator(device=dev).manual_seed(1)
kmax = max(args.k)
acts = torch.rand((kmax, args.n, 3), generator=gen, device=dev) * 2 - 1
for _ in range(max(1, 1500 // kmax)):
    env.step_k(acts, last_only=True)
nw = (args.n + 63) // 64
for K in args.k:
    out = env.step_k(acts[:K])
    for rep in range(3):
        env.stats.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_k(acts[:K], out=out); e1.record(); torch.cuda.synchronize()
        reg = env.stats.cpu().numpy()[nw: 2 * nw]
        st = env.stats.cpu().numpy()[:nw]
        t0, t1, t2, t3, rs = st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4]
        base = t0.min()
        us = lambda x: (x - base) / 100.0
        loop = (t2 - t1) / 100.0
        print(f"K={K:3d} rep {rep}: launch {e0.elapsed_time(e1) * 1e3:7.1f} us by events | wave entry {us(t0).min():.2f}..{us(t0).max():.2f} | loads done "
              f"{us(t1).mean():.2f} (max {us(t1).max():.2f}) | K loop mean {loop.mean():.2f} min {loop.min():.2f} max {loop.max():.2f} us = {loop.mean() / K:.3f} us/step | "
              f"loop end mean {us(t2).mean():.2f} max {us(t2).max():.2f} | wave end mean {us(t3).mean():.2f} max {us(t3).max():.2f} | resets/wave mean {rs.mean():.2f} max {rs.max():.0f} | "
              f"corr(loop, resets) {np.corrcoef(loop, rs)[0, 1]:.2f}; loop us by resets: " +
              " ".join(f"{int(r)}:{loop[rs == r].mean():.1f}" for r in np.unique(rs)[:8]))
        nd = np.maximum(st[:, 4], 1e-9)
        cyc_all = reg[:, :4].sum(0) / (nw * K)
        cyc_done = reg[:, 4:].sum(0) / st[:, 4].sum()
        cyc_plain = (reg[:, :4].sum(0) - reg[:, 4:].sum(0)) / (nw * K - st[:, 4].sum())
        names = ("outputs+obs tile", "terminal outputs", "reset", "obs stores")
        print("      shader cycles per step [region: plain step | step with an episode end]: " +
              "; ".join(f"{n}: {a:.0f} | {b:.0f}" for n, a, b in zip(names, cyc_plain, cyc_done)) + f"; total plain {cyc_plain.sum():.0f} | with end {cyc_done.sum():.0f}")
