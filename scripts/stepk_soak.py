"""Soak of r6_step_k at BASELINE configs[1] (dev tool, GPU box): many K = 16 launches of 65 536 auto-resetting envs (the pair
kernel: bounded waits between the waves of a workgroup), with the state invariants checked every 2000 launches; any abort of a
wait would surface as R6Error from the next call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n, K = 65536, 16
env = BatchedRocket6DOF(n, default_env_config(), integrator="euler", timestep=0.04, seed=11, track_episodes=True)
env.reset()
g = torch.Generator(device=env.device).manual_seed(5)
pool = [torch.rand((K, n, 3), generator=g, device=env.device) * 2 - 1 for _ in range(3)]
t0 = time.time()
for i in range(launches):
    env.step_k(pool[i % 3], last_only=(i % 2 == 0))
    if (i + 1) % 2000 == 0:
        torch.cuda.synchronize()
        st = env.get_state()
        assert bool(torch.isfinite(st).all()), "non-finite state"
        qn = st[:, 6:10].norm(dim=1)
        assert float((qn - 1).abs().max()) < 1e-5
        print(f"{i + 1} launches = {(i + 1) * K * n / 1e9:.1f} G env-steps, {time.time() - t0:.1f} s, episodes {env.episode_stats()['episodes']:.0f}", flush=True)
print("ok")
