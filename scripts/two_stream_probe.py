"""Does the tail of one step launch hide under the ramp of another?  2^20 envs as ONE batch on one stream against the same
envs as TWO (or four) independent half-size batches stepped on separate HIP streams (envs are independent, so a caller
may shard one GPU's batch this way; within one stream consecutive launches serialise).  Dev tool, GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

cfg = default_env_config()
dev = torch.device("cuda:0")
N = 1 << 20
steps = 300
gen = torch.Generator(device=dev).manual_seed(1)


def run(parts):
    n = N // parts
    envs = [BatchedRocket6DOF(n, cfg, integrator="rk4", device=dev, seed=42, env_offset=i * n, record_flags=False,
                              record_terms=False, track_episodes=False) for i in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    pool = [[(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)] for _ in range(parts)]
    torch.cuda.synchronize()
    for i, e in enumerate(envs):
        with torch.cuda.stream(streams[i]):
            e.reset()
            for k in range(2000):
                e.step_async(pool[i][k % 2])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        for s in streams:
            s.wait_event(t0)
        for k in range(steps):
            for i, e in enumerate(envs):
                with torch.cuda.stream(streams[i]):
                    e.step_async(pool[i][k % 2])
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) * 1e3 / steps)
    print(f"{parts} batch(es) of {n} envs on {parts} stream(s): {best:.2f} us per step of {N} envs -> {N / best / 1e3:.2f} G env-steps/s, "
          f"{189 * N / best / 1e6:.0f} GB/s algorithmic")


for parts in (1, 2, 4, 1, 2):
    run(parts)
