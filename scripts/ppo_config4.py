"""PPO iteration at BASELINE configs[4]'s per-GPU size (2^20 envs): rollout / update wall time for the reference's net
(shared trunk [128, 64]) and SB3's default (two separate (64, 64) MLPs).
    python scripts/ppo_config4.py [n_envs] [n_steps] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
for arch, shared in (((128, 64), True), ((64, 64), False)):
    env = BatchedRocket6DOF(n, default_env_config(), remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, seed=1)
    algo = PPO(env, PPOConfig(n_steps=T, batch_size=bs, n_epochs=4, net_arch=arch, shared_trunk=shared, log_std_min=-2.5))
    for _ in range(2):
        algo.collect_rollouts()
        algo.train()
    torch.cuda.synchronize()
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        algo.collect_rollouts()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        algo.train()
        torch.cuda.synchronize()
        t.append((t1 - t0, time.perf_counter() - t1))
    r, u = (sum(x) / len(t) for x in zip(*t))
    steps = 4 * (T * n // bs)
    print(f"n={n} T={T} batch={bs} arch={arch} shared={shared} fused_update={algo.fused_update is not None}: rollout {r * 1e3:.1f} ms, "
          f"update {u * 1e3:.1f} ms ({steps} minibatch steps, {u / steps * 1e3:.3f} ms each) -> {T * n / (r + u) / 1e6:.1f} M steps/s", flush=True)
    del algo, env
    torch.cuda.empty_cache()
