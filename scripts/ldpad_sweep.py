#!/usr/bin/env python3
"""Row-pitch skew of the SoA arrays (R6_LD_PAD, floats) against the step time at 2^20 envs: one process, envs with different pitches
timed interleaved (dev tool, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
dev = torch.device("cuda:0"); n = 1 << 20
pads = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,256,512,1024,1536,2560,3072,5120".split(","))]
gen = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)]
envs = {}
for p in pads:
    os.environ["R6_LD_PAD"] = str(p)
    e = BatchedRocket6DOF(n, default_env_config(), device=dev, record_flags=False, record_terms=False, track_episodes=False, seed=42)
    e.reset()
    for k in range(2000):
        e.step_async(pool[k % 2])
    envs[p] = e
torch.cuda.synchronize()
res = {p: [] for p in pads}
for r in range(11):
    for p in pads:
        e = envs[p]
        for j in range(10):
            e.step_async(pool[j % 2])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for j in range(200):
            e.step_async(pool[j % 2])
        b.record(); torch.cuda.synchronize()
        res[p].append(a.elapsed_time(b) / 200 * 1e3)
for p in pads:
    v = sorted(res[p])
    print(f"ld = n + {p:5d}: median {v[len(v)//2]:6.2f} us  min {v[0]:6.2f}  max {v[-1]:6.2f}")
