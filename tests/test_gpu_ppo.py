"""GPU: the on-device PPO loop drives the fused make_env() stack end to end."""
import math

import pytest

pytestmark = pytest.mark.gpu


def test_ppo_improves_return_on_the_real_env():
    import torch

    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
    from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

    env = BatchedRocket6DOF(8192, default_env_config(), remove_mass=True, clip_reward=(-1, 100),
                            max_episode_steps=1500, seed=42)
    algo = PPO(env, PPOConfig(n_steps=128, batch_size=1 << 16, net_arch=(128, 64), seed=42))   # main_6DOF.py:113
    hist = algo.learn(40_000_000)
    for h in hist:
        assert all(math.isfinite(v) for v in h.values())
    early = min(h["ep_rew_mean"] for h in hist[1:6])
    late = hist[-1]["ep_rew_mean"]
    print(f"[ppo] {len(hist)} iterations, {hist[-1]['timesteps']} steps: ep_rew_mean {early:.2f} -> {late:.2f}, "
          f"ep_len_mean {hist[-1]['ep_len_mean']:.0f}")
    assert late > early + 10.0 and late > -20.0
    assert env.obs.shape == (8192, 13) and bool(torch.isfinite(env.obs).all())
