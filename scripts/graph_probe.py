"""Does a hipGraph of back-to-back r6_step launches shorten the launch-to-launch time at N = 2^20?
(probe only: the baked step_index repeats per replay)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

dev = torch.device("cuda:0")
n = 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), device=dev, auto_reset=True, seed=42, record_flags=False,
                        record_terms=False, track_episodes=False)
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(8)]
env.reset()
for k in range(320):
    env.step_async(pool[k % 8])
torch.cuda.synchronize()


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def eager():
    for k in range(8):
        env.step_async(pool[k])


print(f"stream launches: {timed(eager, 25) / 200:.2f} us/step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eager()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        eager()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print(f"graph replay   : {timed(g.replay, 25) / 200:.2f} us/step")
print(f"stream launches: {timed(eager, 25) / 200:.2f} us/step")
