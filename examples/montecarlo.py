#!/usr/bin/env python3
"""Counterpart of the reference's montecarlo_script.py: fly N episodes with a trained policy (deterministic actions)
and report mean / standard deviation of the terminal-state errors it writes to results_montecarlo.csv
(montecarlo_script.py:33-51: final position / velocity error, attitude error, angular-velocity error, terminal
mass).  The reference runs 30 episodes of one env through SB3's evaluate_policy; here N envs fly at once, the policy
forward and the env step are one HIP launch each.

    python examples/montecarlo.py --checkpoint /path/to/best_model_2bo71j9m.zip --episodes 100000

``--checkpoint`` is a stable-baselines3 PPO zip (only its tensors are read, torch.load(weights_only=True)),
``--policy`` a state_dict saved by examples/train_ppo.py; without either the shipped policy
(examples/policies/landing_policy_shared_128_64.pt) is flown.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_rocket_6dof_amd import BatchedRocket6DOF, load_config
from rl_rocket_6dof_amd.policy import FusedActorCritic, load_policy, load_sb3_policy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default=None, help="stable-baselines3 PPO zip (the reference's best_model_*.zip)")
    ap.add_argument("--policy", default=None, help="state_dict written by examples/train_ppo.py --save")
    ap.add_argument("--episodes", type=int, default=100_000)
    ap.add_argument("--config", default=None)
    ap.add_argument("--integrator", default="rk4", choices=["rk4", "rk45", "euler"])
    ap.add_argument("--max-steps", type=int, default=1500)       # main_6DOF.py:31 max_time / timestep
    args = ap.parse_args()

    _, env_config = load_config(args.config)
    n = args.episodes
    # the full 14-dim observation is kept in the env (the terminal mass is one of the reported columns);
    # the policy sees the RemoveMassFromObs view (wrappers.py:119-126) when it was trained on 13 inputs
    if args.checkpoint:
        policy = load_sb3_policy(args.checkpoint)
    elif args.policy:
        policy = load_policy(args.policy)
    else:   # the policy trained by examples/train_ppo.py that ships with the package
        policy = load_policy(os.path.join(os.path.dirname(os.path.abspath(__file__)), "policies", "landing_policy_shared_128_64.pt"))
    fused = FusedActorCritic(policy)
    env = BatchedRocket6DOF(n, env_config, integrator=args.integrator, auto_reset=True, track_episodes=True,
                            seed=env_config.get("seed", 42))
    obs = env.reset()
    first = torch.ones(n, dtype=torch.bool, device=env.device)           # still flying its first episode
    cols = {}
    for _ in range(args.max_steps):
        pol_in = obs[:, : fused.obs_dim].contiguous()
        obs, _, done, _ = env.step(fused(pol_in, deterministic=True)[0])
        if bool(done.any()):
            te = env.terminal_errors()
            keep = first[te["env_index"]]
            for k, v in te.items():
                if k != "env_index":
                    cols.setdefault(k, []).append(v[keep])
            first &= ~done
        if not bool(first.any()):
            break
    print(f"{n - int(first.sum())} of {n} episodes finished")
    stats = env.episode_stats() if env.stats is not None else None
    if stats and stats["episodes"] > 0:
        print(f"landed (all conditions of rocket_env.py:382-390): {100 * stats['landed'] / stats['episodes']:.2f} % of "
              f"{int(stats['episodes'])} finished episodes; ground contact {100 * stats['ground'] / stats['episodes']:.1f} %")
    for k, v in cols.items():
        v = torch.cat(v)
        print(f"The {k} has mean:{float(v.mean()):.4f} and standard deviation: {float(v.std()):.4f}")


if __name__ == "__main__":
    main()
