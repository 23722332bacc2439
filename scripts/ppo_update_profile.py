"""Kernel-level breakdown of one PPO update epoch at N = 2^20 (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), device="cuda:0", remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, seed=1)
algo = PPO(env, PPOConfig(n_steps=8, batch_size=n, n_epochs=2, net_arch=(64, 64)))
algo.collect_rollouts()
torch.cuda.synchronize()
t0 = time.time()
algo.train()
torch.cuda.synchronize()
t1 = time.time()
algo.train()
torch.cuda.synchronize()
print(f"train(): {(time.time() - t1) * 1e3:.1f} ms for {2 * 8} minibatches of {n} samples -> {(time.time() - t1) / 16 * 1e3:.2f} ms each (first call {(t1 - t0) * 1e3:.0f} ms)")
