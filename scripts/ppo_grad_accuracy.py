"""Accuracy of r6_ppo_minibatch_grad builds against autograd in FLOAT64 (dev tool, GPU box): worst error of a parameter gradient
relative to its tensor's largest entry, next to the same figure for torch's own fp32 autograd.
    python scripts/ppo_grad_accuracy.py [name=path.so ...]"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_rocket_6dof_amd.ppo import ActorCritic, FusedMinibatchGrad

libs = dict(a.split("=") for a in sys.argv[1:] if "=" in a) or {"default": None}
dev = "cuda"


def loss_of(pol, obs, action, old_logp, adv, ret, mb, norm_adv, clip=0.2, vf=0.5, ec=0.01):
    a = adv[mb]
    if norm_adv:
        a = (a - a.mean()) / (a.std() + 1e-8)
    values, logp, entropy = pol.evaluate(obs[mb], action[mb])
    ratio = torch.exp(logp - old_logp[mb])
    pg = -torch.min(a * ratio, a * ratio.clamp(1 - clip, 1 + clip)).mean()
    return pg + ec * (-entropy.mean()) + vf * torch.nn.functional.mse_loss(ret[mb], values)


for arch, shared in (((128, 64), True), ((64, 64), False)):
    for od, n_rows, n_mb, norm_adv in ((13, 5000, 3001, True), (14, 70000, 65536, False), (13, 300, 17, True), (13, 1 << 20, 65536, True)):
        torch.manual_seed(1)
        pol = ActorCritic(od, 3, arch, log_std_init=-0.3, shared_trunk=shared).to(dev)
        with torch.no_grad():
            pol.pi[-1].weight.mul_(30.0)
            pol.pi[-1].bias.uniform_(-0.2, 0.2)
            pol.log_std.copy_(torch.tensor([-0.4, 0.2, -1.0]))
        obs = torch.randn(n_rows, od, device=dev) * 0.7
        with torch.no_grad():
            mean_all, _ = pol._forward(obs)
        action = (mean_all + torch.randn(n_rows, 3, device=dev) * pol.log_std.exp()).detach()
        old_logp = pol._logp(mean_all, action).detach() + torch.randn(n_rows, device=dev) * 0.15
        adv, ret = torch.randn(n_rows, device=dev) * 2 + 0.3, torch.randn(n_rows, device=dev)
        mb = torch.randperm(n_rows, device=dev)[:n_mb]
        pol64 = copy.deepcopy(pol).double()
        g64 = torch.autograd.grad(loss_of(pol64, obs.double(), action.double(), old_logp.double(), adv.double(), ret.double(), mb, norm_adv), list(pol64.parameters()))
        g32 = torch.autograd.grad(loss_of(pol, obs, action, old_logp, adv, ret, mb, norm_adv), list(pol.parameters()))
        names = [n for n, _ in pol.named_parameters()]

        def worst(gs):
            rel = [float((g.double() - r).abs().max()) / max(float(r.abs().max()), 1e-300) for g, r in zip(gs, g64)]
            k = max(range(len(rel)), key=rel.__getitem__)
            return f"{rel[k]:.2e} ({names[k]})"

        line = f"{str(arch):10s} od={od} n={n_mb:6d} norm_adv={int(norm_adv)} | torch fp32 autograd {worst(g32)}"
        for name, path in libs.items():
            fused = FusedMinibatchGrad(pol, lib_path=path)
            norm = fused.adv_norm(adv, mb).clone() if norm_adv else None
            for p in pol.parameters():
                p.grad = None
            fused(obs, action, old_logp, adv, ret, mb, norm, 0.2, 0.5, 0.01)
            torch.cuda.synchronize()
            line += f" | {name} {worst([p.grad for p in pol.parameters()])}"
        print(line, flush=True)
