"""A/B of r6_ppo_minibatch_grad builds in one process (dev tool): us per call, interleaved rounds.
    python scripts/ppo_grad_ab.py name=path.so [name=path.so ...] [--n 65536] [--rows 1048576] [--sep]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_rocket_6dof_amd.ppo import ActorCritic, FusedMinibatchGrad

args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
opt = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
n, rows, sep = int(opt.get("--n", 65536)), int(opt.get("--rows", 1 << 20)), "--sep" in sys.argv
libs = dict(a.split("=") for a in args) or {"default": None}
dev, od = "cuda", 13
torch.manual_seed(0)
obs = torch.randn(rows, od, device=dev)
act = torch.randn(rows, 3, device=dev)
old, adv, ret = (torch.randn(rows, device=dev) for _ in range(3))
mb = torch.randperm(rows, device=dev)[:n].contiguous()
fused = {}
for name, path in libs.items():
    pol = ActorCritic(od, 3, (64, 64) if sep else (128, 64), shared_trunk=not sep).to(dev)
    fused[name] = FusedMinibatchGrad(pol, lib_path=path)
best = {k: [] for k in fused}
for rnd in range(7):
    for name, f in fused.items():
        for _ in range(5):
            f(obs, act, old, adv, ret, mb, None, 0.2, 0.5, 0.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            f(obs, act, old, adv, ret, mb, None, 0.2, 0.5, 0.0)
        e1.record()
        torch.cuda.synchronize()
        best[name].append(e0.elapsed_time(e1) / 40 * 1e3)
for name, v in best.items():
    v = sorted(v)
    print(f"{name:12s} n={n} rows={rows} {'two (64,64) MLPs' if sep else 'shared [128,64]'}: gradient + reduce median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f} us")
