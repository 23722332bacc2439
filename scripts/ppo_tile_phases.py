"""Per-phase time inside one 32-sample tile of ppo_grad_kernel (wave 0 of workgroup 0), from an R6U_STAMPS build:
    scripts/build_variant.sh stamps -DR6U_STAMPS && python scripts/ppo_tile_phases.py
The stamps go into the free half of the partial-sum scratch (128 workgroups at 16384 samples)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_rocket_6dof_amd.ppo import ActorCritic, FusedMinibatchGrad

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl_rocket_6dof_amd", "csrc", "variants", "libr6dof_stamps.so")
dev = "cuda"
torch.manual_seed(0)
od, nmb, rows = 13, 16384, 1 << 18
pol = ActorCritic(od, 3, (128, 64), shared_trunk=True).to(dev)
obs = torch.randn(rows, od, device=dev)
act = torch.randn(rows, 3, device=dev)
old, adv, ret = (torch.randn(rows, device=dev) for _ in range(3))
mb = torch.randperm(rows, device=dev)[:nmb]
f = FusedMinibatchGrad(pol, lib_path=LIB)
nw, reps = 128, 20
P = f.scratch.numel() // 256   # floats of one workgroup's partial set (r6_ppo_scratch_floats() / 256); its last 16 are the scalars
f.scratch.zero_()
for _ in range(reps):
    f(obs, act, old, adv, ret, mb, None, 0.2, 0.5, 0.0)
torch.cuda.synchronize()
st = f.scratch[nw * P:nw * P + 8].cpu() / reps / 100.0
names = ["F1", "F2", "heads+loss", "G3+staging", "delta2+db2", "G2 x4", "B2 x4", "G1 x4"]
co = f.scratch[:nw * P].view(nw, P)[:, P - 4:P - 1].cpu() / 100.0   # last launch: per workgroup
print("us per launch (mean over workgroups, wave 0): image staging %.1f, tiles %.1f, workgroup reduction + write %.1f" % tuple(co.mean(0).tolist()))
print("us per tile:", ", ".join(f"{n} {v:.2f}" for n, v in zip(names, st.tolist())), "| sum %.1f" % float(st.sum()))
