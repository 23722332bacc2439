#!/usr/bin/env python3
"""The fused closed-loop rollout (r6_rollout) against the launch sequence it replaces (dev tool, GPU box): ms per rollout of
T steps at several env counts, both SB3 topologies, the reference's make_env() stack."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
from rl_rocket_6dof_amd.policy import FusedActorCritic
from rl_rocket_6dof_amd.ppo import ActorCritic

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="+", default=[16384, 65536, 1 << 18, 1 << 20])
ap.add_argument("--T", type=int, default=64)
ap.add_argument("--reps", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = default_env_config()
for arch, shared in (((128, 64), True), ((64, 64), False)):
    pol = ActorCritic(13, 3, arch, log_std_init=-0.5, shared_trunk=shared).to(dev)
    fused = FusedActorCritic(pol, seed=1)
    for n in args.n:
        T = args.T if n <= (1 << 18) else 16
        mk = lambda: BatchedRocket6DOF(n, cfg, device=dev, seed=3, remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, record_flags=True)
        f32 = dict(dtype=torch.float32, device=dev)
        buf = {"obs_seq": torch.zeros((T + 1, n, 13), **f32), "raw_action": torch.zeros((T, n, 3), **f32), "value": torch.zeros((T, n), **f32),
               "logp": torch.zeros((T, n), **f32), "reward": torch.zeros((T, n), **f32), "done": torch.zeros((T, n), dtype=torch.uint8, device=dev),
               "flags": torch.zeros((T, n), dtype=torch.uint8, device=dev), "boot_value": torch.zeros((T, n), **f32)}
        act = torch.zeros((n, 3), **f32)
        scr = (torch.zeros((n, 3), **f32), torch.zeros((n, 3), **f32), torch.zeros((n,), **f32))

        def seq(env, s0):
            for j in range(T):
                fused(buf["obs_seq"][j], step_index=s0 + j, out=(act, buf["raw_action"][j], buf["value"][j], buf["logp"][j]))
                env.step_into(act, obs=buf["obs_seq"][j + 1], reward=buf["reward"][j], done=buf["done"][j], flags=buf["flags"][j])
                fused(env.final_obs, deterministic=True, step_index=0, out=(scr[0], scr[1], buf["boot_value"][j], scr[2]),
                      row_flags=buf["flags"][j], flag_mask=4)

        res = {}
        for name in ("sequence (eager)", "sequence (graph)", "fused r6_rollout"):
            env = mk()
            env.reset()
            buf["obs_seq"][0].copy_(env.obs)
            run = (lambda s0: env.rollout(fused, T, policy_step_index=s0, **buf)) if name.startswith("fused") else (lambda s0: seq(env, s0))
            for w in range(3):
                run(1 + w * T)
            torch.cuda.synchronize()
            if "graph" in name:
                side = torch.cuda.Stream(device=dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    run(1000)
                run = lambda s0: g.replay()
            best = 1e9
            for r in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(5000 + r * T); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            res[name] = best
            del env
        print(f"arch {arch}{' shared' if shared else ''} n={n:8d} T={T}: " + "  ".join(f"{k} {v:7.3f} ms ({n * T / v * 1e-6:7.1f} G env-steps/s)" for k, v in res.items()), flush=True)
