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
st = f.scratch[nw * P:nw * P + 20].cpu() / 100.0      # us since the tile's first stamp
names = ["start", "forward", "heads+loss", "dW3 sums", "delta2+db2"] + [f"{p}[{ib}]" for ib in range(4) for p in ("G2", "B2", "G1")]
prev = 0.0
out = []
for n_, v in zip(names[1:], st.tolist()[1:17]):
    out.append(f"{n_} {v - prev:.2f}")
    prev = v
print("us per phase of wave 0's last tile (an R6U_STAMPS build spills: relative sizes only):", ", ".join(out), "| tile %.1f" % prev)
