"""Where the fixed cost of ppo_grad_kernel goes (R6U_STAMPS build): per workgroup the time spent staging the weight images,
in the tile loop and in the end-of-launch reduction, plus the tile phases of wave 0 (dev tool, GPU box):
    scripts/build_variant.sh stamps -DR6U_STAMPS && python scripts/ppo_grad_fixed_cost.py [--n=65536] [--sep]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_rocket_6dof_amd.ppo import ActorCritic, FusedMinibatchGrad

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl_rocket_6dof_amd", "csrc", "variants", "libr6dof_stamps.so")
opt = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
n, sep, sset = int(opt.get("--n", 65536)), "--sep" in sys.argv, int(opt.get("--set", 0))
if sset:
    LIB = LIB.replace("stamps.so", f"stamps{sset}.so")     # scripts/build_variant.sh stamps1 -DR6U_STAMPS -DR6U_STAMP_SET=1
dev, od, rows = "cuda", 13, 1 << 20
torch.manual_seed(0)
pol = ActorCritic(od, 3, (64, 64) if sep else (128, 64), shared_trunk=not sep).to(dev)
obs = torch.randn(rows, od, device=dev)
act = torch.randn(rows, 3, device=dev)
old, adv, ret = (torch.randn(rows, device=dev) for _ in range(3))
mb = torch.randperm(rows, device=dev)[:n].contiguous()
f = FusedMinibatchGrad(pol, lib_path=LIB)
for _ in range(20):
    f(obs, act, old, adv, ret, mb, None, 0.2, 0.5, 0.0)
torch.cuda.synchronize()
P = f.scratch.numel() // 256
nw = min(256, (n + 127) // 128)
tab = f.scratch[: nw * P].reshape(nw, P)[:, P - 16 + 12: P - 16 + (16 if sset else 15)].cpu() / 100.0     # us: staging, tile loop, reduction
NAMES = {0: ("staging", "tile loop", "reduction"), 1: ("st: params issued", "st: tile issued", "st: arrived", "st: lds+barrier"), 2: ("rd: barrier", "rd: xch wr", "rd: barrier", "rd: sum+st")}[sset]
for name, colv in zip(NAMES, tab.t()):
    print(f"{name:18s} us per workgroup: median {colv.median():6.2f}  min {colv.min():6.2f}  max {colv.max():6.2f}")
print(f"sum of the medians {float(tab.median(0).values.sum()):.1f} us over {nw} workgroups, n = {n}")
