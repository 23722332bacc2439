#!/usr/bin/env python3
"""GPU time of one PPO minibatch step (dev tool): the fused gradient launch against the torch path (autograd +
fused loss kernel), same minibatch, 65 536 samples of a 2^20-row rollout buffer; then the whole step incl. clip + Adam."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd.ppo import ActorCritic, FusedMinibatchGrad, _FusedPPOLoss

dev = "cuda"
torch.manual_seed(0)
rows, nmb, od = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16, 13
pol = ActorCritic(od, 3, (128, 64), log_std_init=0.0, shared_trunk=True).to(dev)
obs = torch.randn(rows, od, device=dev)
with torch.no_grad():
    mean, _ = pol._forward(obs)
act = (mean + torch.randn(rows, 3, device=dev)).detach()
old = pol._logp(mean, act).detach()
adv, ret = torch.randn(rows, device=dev), torch.randn(rows, device=dev)
mb = torch.randperm(rows, device=dev)[:nmb]
fused = FusedMinibatchGrad(pol)
opt = torch.optim.Adam(pol.parameters(), lr=0.0, eps=1e-5)


def t_fused():
    fused(obs, act, old, adv, ret, mb, fused.adv_norm(adv, mb), 0.2, 0.5, 0.0)


def t_torch():
    a = adv[mb]
    std, m = torch.std_mean(a)
    mean, values = pol._forward(obs[mb])
    loss, st = _FusedPPOLoss.apply(mean, values, pol.log_std, act[mb], old[mb], a, ret[mb], torch.stack([m, 1.0 / (std + 1e-8)]), 0.2, 0.5, 0.0)
    opt.zero_grad(set_to_none=True)
    loss.backward()


def step(fn):
    fn()
    torch.nn.utils.clip_grad_norm_(pol.parameters(), 0.5)
    opt.step()


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"minibatch of {nmb} samples: gradient  fused {timeit(t_fused):8.1f} us   torch {timeit(t_torch):8.1f} us")
print(f"                      whole step  fused {timeit(lambda: step(t_fused)):8.1f} us   torch {timeit(lambda: step(t_torch)):8.1f} us   (eager launches)")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    t_fused()
print(f"                      fused gradient replayed as a graph {timeit(g.replay):8.1f} us")
lr = torch.tensor(0.0, device=dev)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    t_fused()
    fused.adam_step(lr, 0.5, -2.5)
print(f"                      fused gradient + r6_adam_clip_step, as a graph {timeit(g2.replay):8.1f} us   (the whole minibatch step of ppo.PPO)")
