#!/usr/bin/env python3
"""BASELINE configs[1] through r6_step_k, nothing else (the program scripts/pmc_stepk.sh profiles): 65 536 envs, Euler
dt 0.04, K env steps per launch, all per-step outputs stored."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--k", type=int, default=16)
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--integrator", default="euler")
ap.add_argument("--dt", type=float, default=0.04)
args = ap.parse_args()
dev = torch.device("cuda:0")
env = BatchedRocket6DOF(args.n, default_env_config(), device=dev, integrator=args.integrator, timestep=args.dt, record_flags=False,
                        track_episodes=False, seed=42)
env.reset()
gen = torch.Generator(device=dev).manual_seed(1)
acts = torch.rand((args.k, args.n, 3), generator=gen, device=dev) * 2 - 1
for _ in range(max(1, 1500 // args.k)):
    env.step_k(acts, last_only=True)
out = env.step_k(acts)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    env.step_k(acts, out=out)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e3 / (args.reps * args.k)
print(f"n={args.n} K={args.k} {args.integrator} dt={args.dt}: {t:.3f} us per env-step, {args.n / t * 1e-3:.2f} G env-steps/s")
