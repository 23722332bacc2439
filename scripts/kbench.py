#!/usr/bin/env python3
"""Kernel micro-benchmark (dev tool, GPU box): interleaved rounds of several step-kernel
configurations in ONE process; reports median / min launch-to-launch time per step."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--configs", default="rk4,rk4_noreset,euler,rk4_wrapped")
ap.add_argument("--libs", default="", help="comma list of name=path kernel builds to A/B (default: in-tree)")
ap.add_argument("--burnin", type=int, default=400, help="auto-reset steps before timing (desynchronises episodes)")
args = ap.parse_args()
cfg = default_env_config()
dev = torch.device("cuda:0")
n = args.n

libs = dict(x.split("=") for x in args.libs.split(",")) if args.libs else {"": None}

def make(name):
    name, _, libname = name.partition("@")
    kw = dict(device=dev, record_flags=False, record_terms=False, track_episodes=False, seed=42, lib_path=libs.get(libname))
    if name == "rk4": return BatchedRocket6DOF(n, cfg, integrator="rk4", **kw)
    if name == "rk4_noreset": return BatchedRocket6DOF(n, cfg, integrator="rk4", auto_reset=False, **kw)
    if name == "rk45": return BatchedRocket6DOF(n, cfg, integrator="rk45", **kw)
    if name == "euler": return BatchedRocket6DOF(n, cfg, integrator="euler", **kw)
    if name == "euler_dt004": return BatchedRocket6DOF(n, cfg, integrator="euler", timestep=0.04, **kw)
    if name == "rk4_wrapped":
        kw.update(track_episodes=True)
        return BatchedRocket6DOF(n, cfg, integrator="rk4", remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, **kw)
    if name == "rk4_flags_terms":
        kw.update(record_flags=True, record_terms=True)
        return BatchedRocket6DOF(n, cfg, integrator="rk4", **kw)
    raise SystemExit(name)

names = [f"{c}@{l}" if l else c for c in args.configs.split(",") for l in libs]
envs = {k: make(k) for k in names}
gen = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(4)]
for e in envs.values():
    e.reset()
    for k in range(200 if not e.auto_reset else args.burnin):   # reach a steady mix of flight phases
        e.step_async(pool[k % 4])
        if not e.auto_reset and k % 50 == 49:
            e.reset(mask=e.done)
torch.cuda.synchronize()
res = {k: [] for k in names}
for r in range(args.rounds):
    for k in names:
        e = envs[k]
        if not e.auto_reset:
            e.reset()
            for j in range(20): e.step_async(pool[j % 4])
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for j in range(args.steps):
            e.step_async(pool[j % 4])
        ev1.record()
        torch.cuda.synchronize()
        res[k].append(ev0.elapsed_time(ev1) / args.steps * 1e3)
for k in names:
    v = sorted(res[k])
    b = 201 if "wrapped" in k else 189
    med = v[len(v) // 2]
    print(f"{k:28s} n={n} us/step median {med:8.2f} min {v[0]:8.2f}  -> {n/med*1e-3:8.2f} G env-steps/s  alg {b*n/med*1e-3:8.1f} GB/s ({b*n/med*1e-3/80:.1f}% of 8 TB/s)")
