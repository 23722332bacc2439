"""Long random-action soak of the step kernel (dev tool, GPU box): K steps of 2^20 auto-resetting envs, with the
state invariants (finite, unit quaternion, inside the termination box, mass within its bounds) and the episode
accounting checked every K/10 steps."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

K = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
integ = sys.argv[2] if len(sys.argv) > 2 else "rk4"
n = 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), integrator=integ, seed=123, track_episodes=True)
env.reset()
g = torch.Generator(device=env.device).manual_seed(5)
pool = [torch.rand((n, 3), generator=g, device=env.device) * 2 - 1 for _ in range(7)]
t0 = time.time()
done_total = 0.0
for k in range(K):
    env.step_async(pool[k % 7])
    if (k + 1) % (K // 10) == 0:
        st = env.get_state()
        assert bool(torch.isfinite(st).all()), "non-finite state"
        qn = st[:, 6:10].norm(dim=1)
        assert float((qn - 1).abs().max()) < 1e-5, float((qn - 1).abs().max())
        assert float(st[:, 0].min()) > -31.0 and float(st[:, 0].max()) < 2161.0
        assert float(st[:, 13].min()) > 0.0
        s = env.episode_stats()
        done_total += s["episodes"]
        assert s["ground"] + s["bounds"] + s.get("truncated", 0) >= s["episodes"] > 0
        print(f"step {k + 1}: ok, {s['episodes']:.0f} episodes in this window (ground {s['ground']:.0f}, bounds {s['bounds']:.0f}, "
              f"landed {s['landed']:.0f}), mean length {s['length_sum'] / s['episodes']:.1f}, {time.time() - t0:.1f} s", flush=True)
print(f"soak ok: {K} steps x {n} envs = {K * n:.3e} env-steps, {done_total:.3e} episodes, {time.time() - t0:.1f} s wall")
