#!/usr/bin/env python3
"""r6_step_k against one launch per step (dev tool, GPU box): us per env-step for several (n, K)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="+", default=[65536, 1 << 20])
ap.add_argument("--k", type=int, nargs="+", default=[1, 4, 16, 64])
ap.add_argument("--integrator", default="euler")
ap.add_argument("--dt", type=float, default=0.04)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--lib", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = default_env_config()
for n in args.n:
    env = BatchedRocket6DOF(n, cfg, device=dev, integrator=args.integrator, timestep=args.dt, record_flags=False,
                            track_episodes=False, seed=42, lib_path=args.lib)
    env.reset()
    gen = torch.Generator(device=dev).manual_seed(1)
    kmax = max(args.k)
    acts = (torch.rand((kmax, n, 3), generator=gen, device=dev) * 2 - 1)
    for _ in range(max(1, 1500 // kmax)):
        env.step_k(acts, last_only=True)
    torch.cuda.synchronize()
    def timed(fn, steps):
        best = 1e9
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
        return best
    t1 = timed(lambda: [env.step_async(acts[j % kmax]) for j in range(256)], 256)
    print(f"n={n:8d} {args.integrator} dt={args.dt}: one launch per step {t1:7.2f} us/step -> {n/t1*1e-3:7.2f} G env-steps/s")
    for K in args.k:
        out = env.step_k(acts[:K])
        reps = max(1, 256 // K)
        t = timed(lambda: [env.step_k(acts[:K], out=out) for _ in range(reps)], reps * K)
        tl = timed(lambda: [env.step_k(acts[:K], last_only=True) for _ in range(reps)], reps * K)
        b_all = 12 + 4 * env.obs_dim + 5 + (2 * 56 + 2 * 8) / K          # action, obs, reward, done + state/epconst read+write / K
        print(f"   K={K:3d}: all outputs {t:7.2f} us/step -> {n/t*1e-3:7.2f} G env-steps/s ({b_all*n/t*1e-3:7.1f} GB/s alg); last-only {tl:7.2f} us/step -> {n/tl*1e-3:7.2f} G env-steps/s")
    del env, acts
