"""Kernel-level breakdown of the PPO minibatch step at the training configuration of examples/train_ppo.py (shared trunk
[128, 64], 65 536-sample minibatches), eager (no graph replay) so that rocprofv3 --kernel-trace --stats sees plain launches."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

n = 16384
env = BatchedRocket6DOF(n, default_env_config(), remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, seed=1)
algo = PPO(env, PPOConfig(n_steps=64, batch_size=1 << 16, n_epochs=2, net_arch=(128, 64), shared_trunk=True, log_std_min=-2.5,
                          graph_update=False))
algo.collect_rollouts()
algo.train()
torch.cuda.synchronize()
t0 = time.time()
algo.train()
torch.cuda.synchronize()
print(f"eager train(): {(time.time() - t0) / 32 * 1e3:.3f} ms per minibatch step of 65536 samples")
