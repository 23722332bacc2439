"""Wall time of the phases of one PPO iteration (rollout / GAE / update) at the training configuration of
examples/train_ppo.py (16 k envs x 64 steps, shared trunk [128, 64], 10 epochs x 16 minibatches of 65 536)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
from rl_rocket_6dof_amd.ppo import PPO, PPOConfig, compute_gae

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = BatchedRocket6DOF(n, default_env_config(), remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, seed=1)
algo = PPO(env, PPOConfig(n_steps=64, batch_size=1 << 16, net_arch=(128, 64), shared_trunk=True, log_std_min=-2.5))
for _ in range(3):
    algo.collect_rollouts()
    algo.train()
torch.cuda.synchronize()
t = []
for _ in range(5):
    t0 = time.perf_counter()
    algo.collect_rollouts()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    algo.train()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t.append((t1 - t0, t2 - t1))
r = sum(a for a, _ in t) / len(t)
u = sum(b for _, b in t) / len(t)
print(f"n={n}: rollout (64 steps + GAE) {r * 1e3:.1f} ms, update (160 minibatch steps) {u * 1e3:.1f} ms = {u / 160 * 1e3:.3f} ms each "
      f"-> {64 * n / (r + u) / 1e6:.1f} M steps/s")
