#!/usr/bin/env python3
"""Attempts per lane and per 64-env tile of the adaptive integrator in the bench's steady state (dev tool, GPU box): what a wave that
follows its slowest lane pays, and what is left after k attempts."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
dev = torch.device("cuda:0")
n = 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), device=dev, integrator="rk45", record_flags=False, record_terms=False, track_episodes=False, record_nfev=True, seed=42)
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)]
env.reset()
for k in range(600):
    env.step_async(pool[k % 2])
out = {}
for rep in range(3):
    for k in range(50):
        env.step_async(pool[k % 2])
    torch.cuda.synchronize()
    nf = env.nfev.cpu().numpy().astype(int)
    att = (nf - 2) // 6
    t = att.reshape(-1, 64)
    tmax = t.max(1)
    rec = {"attempts_per_lane_hist": {int(k): float((att == k).mean()) for k in range(0, 8)}, "mean_attempts_per_lane": float(att.mean()),
           "mean_attempts_per_tile (max over lanes)": float(tmax.mean()),
           "tile_max_hist": {int(k): float((tmax == k).mean()) for k in range(0, 10)}}
    for k in (1, 2, 3):
        left = (t > k).sum(1)
        rec[f"lanes_left_after_{k}_attempts_hist"] = {"0": float((left == 0).mean()), "1-2": float(((left >= 1) & (left <= 2)).mean()), "3-6": float(((left >= 3) & (left <= 6)).mean()),
                                                      "7-16": float(((left >= 7) & (left <= 16)).mean()), "17-32": float(((left >= 17) & (left <= 32)).mean()), "33-64": float((left >= 33).mean()),
                                                      "mean": float(left.mean())}
    out[f"rep{rep}"] = rec
print(json.dumps(out, indent=1))
