#!/usr/bin/env python3
"""Counterpart of the reference's main_6DOF.py:start_training (without wandb / xvfb): PPO on the fused
make_env() stack (RemoveMassFromObs + ClipReward(-1,100) + TimeLimit(max_time/timestep) + Monitor), every
environment on the GPU.

    python examples/train_ppo.py --envs 65536 --timesteps 2e8 [--config config.yaml]
    torchrun --nproc-per-node 8 examples/train_ppo.py --envs 8388608      # env-sharded, RCCL grad all-reduce
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_rocket_6dof_amd import dist as r6dist
from rl_rocket_6dof_amd import load_config
from rl_rocket_6dof_amd.ppo import PPO, PPOConfig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 16, help="total environments (sharded over the ranks)")
    ap.add_argument("--timesteps", type=float, default=5e7)
    ap.add_argument("--n-steps", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=1 << 16)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--lr-final", type=float, default=None, help="linear learning-rate schedule down to this value")
    ap.add_argument("--log-std-min", type=float, default=-2.5,
                    help="floor of the policy's log-std (std >= 0.08); SB3 has none -- pass 'nan' to disable -- but with "
                         "~10^6 samples per update the std collapses (log-std sum -8 after 2e9 steps) and the late phase "
                         "of training turns erratic; with the floor every run so far ends above 99 %% landings")
    ap.add_argument("--net-arch", default="128,64", help="main_6DOF.py:113 uses [128, 64]; SB3 default is 64,64")
    ap.add_argument("--separate", action="store_true",
                    help="two separate MLPs instead of SB3 1.6's meaning of a plain-list net_arch (one shared trunk, "
                         "which is what the reference's net_arch=[128, 64] builds)")
    ap.add_argument("--config", default=None)
    ap.add_argument("--save", default=None, help="write the policy state_dict here (torch.save) whenever the mean episode "
                                                 "return of an iteration is the best so far (the reference keeps its "
                                                 "best_model through SB3's EvalCallback, main_6DOF.py)")
    args = ap.parse_args()

    # R6_DIST_BACKEND=gloo + R6_FORCE_DEVICE=0 rehearse several ranks on ONE GPU (RCCL wants one device per rank)
    rank, world, local = r6dist.init_from_env(backend=os.environ.get("R6_DIST_BACKEND"))
    local = int(os.environ.get("R6_FORCE_DEVICE", local))
    torch.cuda.set_device(local)
    sb3_config, env_config = load_config(args.config)
    max_steps = int(sb3_config["max_time"] / env_config["timestep"])          # main_6DOF.py:31
    env = r6dist.make_sharded_env(args.envs, env_config, device=f"cuda:{local}", remove_mass=True,
                                  clip_reward=(-1, 100), max_episode_steps=max_steps, seed=env_config["seed"])
    cfg = PPOConfig(n_steps=args.n_steps, batch_size=args.batch_size, n_epochs=args.epochs, learning_rate=args.lr, final_learning_rate=args.lr_final, log_std_min=(None if args.log_std_min != args.log_std_min else args.log_std_min),
                    net_arch=tuple(int(x) for x in args.net_arch.split(",")), shared_trunk=not args.separate,
                    seed=env_config["seed"])
    algo = PPO(env, cfg)
    t0 = time.time()

    best = [-1e30]

    def log(info):
        if rank == 0 and args.save and info.get("episodes", 0) > 100 and info["ep_rew_mean"] > best[0]:
            best[0] = info["ep_rew_mean"]
            torch.save(algo.policy.state_dict(), args.save)
            info = dict(info, saved=True)
        if rank == 0:
            info = dict(info, wall_s=round(time.time() - t0, 1), sps=round(info["timesteps"] / (time.time() - t0)))
            print(json.dumps({k: (round(v, 5) if isinstance(v, float) else v) for k, v in info.items()}), flush=True)

    algo.learn(int(args.timesteps), log=log)


if __name__ == "__main__":
    main()
