"""Host-side cost of one BatchedRocket6DOF.step_async call (python + ctypes + hipLaunchKernel), measured
with a batch so small that the GPU is never the bottleneck, next to a bare lib.r6_step loop."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

dev = torch.device("cuda:0")
env = BatchedRocket6DOF(256, default_env_config(), device=dev, auto_reset=True, seed=1)
env.reset()
a = (torch.rand((256, 3), device=dev) * 2 - 1).contiguous()
for _ in range(200):
    env.step_async(a)
torch.cuda.synchronize()
K = 5000
t0 = time.perf_counter()
for _ in range(K):
    env.step_async(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"step_async: {(t1 - t0) / K * 1e6:.2f} us/call issue, {(t2 - t0) / K * 1e6:.2f} us/call incl. drain")
lib, p, b, s = env.lib, C.byref(env.params), C.byref(env._buf), env._stream()
t0 = time.perf_counter()
for k in range(K):
    lib.r6_step(p, b, 1, 100000 + k, 0, s)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"bare r6_step: {(t1 - t0) / K * 1e6:.2f} us/call issue, {(t2 - t0) / K * 1e6:.2f} us/call incl. drain")
