#!/usr/bin/env python3
"""How steady is the 'steady state' of the bench workload?  us per step in consecutive chunks of the same launches, next to
the share of envs that finished in the chunk's last step (dev tool, GPU box)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--chunk", type=int, default=100)
ap.add_argument("--chunks", type=int, default=80)
ap.add_argument("--lib", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
env = BatchedRocket6DOF(args.n, default_env_config(), device=dev, record_flags=False, record_terms=False, track_episodes=False, seed=42, lib_path=args.lib)
gen = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand((args.n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)]
env.reset()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.chunks + 1)]
dones = torch.zeros(args.chunks, device=dev)
torch.cuda.synchronize()
evs[0].record()
for c in range(args.chunks):
    for j in range(args.chunk):
        env.step_async(pool[j % 2])
    dones[c] = env.done.sum()
    evs[c + 1].record()
torch.cuda.synchronize()
d = dones.cpu().numpy() / args.n
for c in range(args.chunks):
    print(f"steps {c * args.chunk:6d}..{(c + 1) * args.chunk:6d}: {evs[c].elapsed_time(evs[c + 1]) / args.chunk * 1e3:7.2f} us/step  done share {d[c] * 100:5.2f} %")
