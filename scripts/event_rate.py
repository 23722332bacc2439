"""Steady-state episode-end statistics of the bench workload: finished envs per step and the share of
64-env wave tiles that contain at least one ground event (those run the event pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

dev = torch.device("cuda:0")
n = 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), device=dev, auto_reset=True, seed=42, record_flags=True)
gen = torch.Generator(device=dev).manual_seed(1234)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(P)]
env.reset()
for k in range(2000):
    env.step_async(pool[k % P])
tot_done = tot_ground = tot_tiles = 0
for k in range(50):
    env.step_async(pool[k % P])
    f = env.flags.clone()
    d = env.done.clone()
    tot_done += int(d.sum())
    g = (f & 1) != 0
    tot_ground += int(g.sum())
    tot_tiles += int(g.view(-1, 64).any(dim=1).sum())
print(f"done/step {tot_done / 50:.0f} ({tot_done / 50 / n * 100:.3f} %), ground events/step {tot_ground / 50:.0f}, "
      f"wave tiles with an event {tot_tiles / 50:.0f} of {n // 64} ({tot_tiles / 50 / (n // 64) * 100:.1f} %)")
