#!/usr/bin/env python3
"""Steps the headline batch with the adaptive integrator (for rocprofv3 passes): python3 scripts/rk45_driver.py <defer> [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config, _lib
defer = int(sys.argv[1]) if len(sys.argv) > 1 else 6
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = 1 << 20
dev = torch.device("cuda:0")
env = BatchedRocket6DOF(n, default_env_config(), device=dev, integrator="rk45", record_flags=False, record_terms=False, track_episodes=False, seed=42)
_lib.tune("rk45_defer", defer)
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)]
env.reset()
for k in range(300 + launches):
    env.step_async(pool[k % 2])
torch.cuda.synchronize()
