"""Closed-loop rollout rate (policy forward + Rocket6DOF.step per step) at N envs on one GPU: torch policy
modules vs the fused HIP forward.  This is the inner loop of PPO.collect_rollouts without the buffer copies."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
from rl_rocket_6dof_amd.policy import FusedActorCritic
from rl_rocket_6dof_amd.ppo import ActorCritic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
for arch in ((64, 64), (128, 64)):
    pol = ActorCritic(14, 3, arch).to(dev)
    for mode in ("torch", "fused"):
        env = BatchedRocket6DOF(n, default_env_config(), device=dev, auto_reset=True, seed=42)
        obs = env.reset()
        fused = FusedActorCritic(pol, seed=1)
        gen = torch.Generator(device=dev).manual_seed(0)
        f32 = dict(dtype=torch.float32, device=dev)
        out = (torch.empty((n, 3), **f32), torch.empty((n, 3), **f32), torch.empty((n,), **f32), torch.empty((n,), **f32))

        def step():
            if mode == "fused":
                a = fused(env.obs, out=out)[0]
            else:
                with torch.no_grad():
                    a = pol.act(env.obs, generator=gen)[0].clamp(-1, 1)
            env.step_async(a)

        for _ in range(20):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / steps * 1e3
        print(f"arch {arch} n={n} {mode:5s} policy: {us:8.1f} us per closed-loop step -> {n / us * 1e-3:6.2f} G env-steps/s")

# Two env groups per GPU, each with its own policy call, on two HIP streams (the pattern of asynchronous vector envs):
# one group's policy forward / launch ramp runs under the other group's env step.
for arch, shared in (((64, 64), False), ((128, 64), True)):
    pol = ActorCritic(14, 3, arch, shared_trunk=shared).to(dev)
    for groups in (1, 2):
        m = n // groups
        envs = [BatchedRocket6DOF(m, default_env_config(), device=dev, auto_reset=True, seed=42, env_offset=g * m) for g in range(groups)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(groups)]
        fused = [FusedActorCritic(pol, seed=1, env_offset=g * m) for g in range(groups)]
        f32 = dict(dtype=torch.float32, device=dev)
        outs = [(torch.empty((m, 3), **f32), torch.empty((m, 3), **f32), torch.empty((m,), **f32), torch.empty((m,), **f32)) for _ in range(groups)]
        for e in envs:
            e.reset()
        torch.cuda.synchronize()

        def step_all():
            for g in range(groups):
                with torch.cuda.stream(streams[g]):
                    envs[g].step_async(fused[g](envs[g].obs, out=outs[g])[0])

        for _ in range(20):
            step_all()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams:
            s.wait_event(e0)
        for _ in range(steps):
            step_all()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / steps * 1e3
        print(f"arch {arch} shared={shared} n={n} fused policy, {groups} group(s) on {groups} stream(s): {us:8.1f} us per closed-loop step "
              f"-> {n / us * 1e-3:6.2f} G env-steps/s")
