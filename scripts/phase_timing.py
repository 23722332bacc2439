"""Where a steady-state r6_step launch spends its time, per workgroup (dev tool; needs the R6_PHASE_TIMING
build: hipcc ... -DR6_PHASE_TIMING -o variants/phase.so).  Stamps: kernel entry, end of the tile loop per
wave, after the workgroup barrier, after the reset pass."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

integ = sys.argv[1] if len(sys.argv) > 1 else "rk4"
dev = torch.device("cuda:0")
n = 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), device=dev, auto_reset=True, seed=42, integrator=integ,
                        track_episodes=True, lib_path=os.path.join(ROOT, "rl_rocket_6dof_amd/csrc/variants/phase.so"))
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(8)]
env.reset()
for k in range(400):
    env.step_async(pool[k % 8])
torch.cuda.synchronize()
rows = []
for k in range(10):
    env.stats.zero_()
    env.step_async(pool[k % 8])
    torch.cuda.synchronize()
    s = env.stats.cpu().numpy().astype(np.int64) & 0xFFFFFFFF   # stamps are stored as exact integers in float64
    s = s[s[:, 0] != 0]
    t0 = s[:, 0].min()
    us = lambda col: (s[:, col] - t0) / 100.0
    wave_end = np.stack([us(1), us(4), us(5), us(6)], axis=1)
    rows.append([len(s), us(0).max(), wave_end.min(), wave_end.mean(), wave_end.max(), us(2).mean(), us(2).max(),
                 us(3).mean(), us(3).max(), s[:, 7].mean(), s[:, 7].max()])
    if k == 9:
        blk = np.nonzero(env.stats.cpu().numpy()[:, 0] != 0)[0]
        n_tiles, nw = n // 64, len(blk) * 4
        heavy = blk * 4 < n_tiles - (n_tiles // nw) * nw        # workgroups whose waves own one tile more
        for name, m in (("heavy", heavy), ("light", ~heavy)):
            if m.any():
                w = wave_end[m]
                print(f"   {name}: {m.sum()} blocks, wave loop end min {w.min():.2f} p10 {np.percentile(w, 10):.2f} "
                      f"median {np.median(w):.2f} p90 {np.percentile(w, 90):.2f} max {w.max():.2f}")
        be = wave_end.max(axis=1)
        print("   block end by blockIdx % 8 :", " ".join(f"{be[blk % 8 == x].mean():.1f}" for x in range(8)))
        print("   block end by blockIdx // (G/8):", " ".join(f"{c.mean():.1f}" for c in np.array_split(be, 8)))
        print("   wave end by wave id:", " ".join(f"{wave_end[:, x].mean():.1f}" for x in range(4)))
        print("   block entry by blockIdx // (G/8):", " ".join(f"{c.mean():.2f}" for c in np.array_split(us(0), 8)))
r = np.array(rows)[2:].mean(axis=0)
print(f"[{integ}] blocks {r[0]:.0f}; last block entry +{r[1]:.2f} us; tile loop ends: first wave {r[2]:.2f}, mean {r[3]:.2f}, "
      f"last {r[4]:.2f}; after barrier mean {r[5]:.2f} max {r[6]:.2f}; after reset pass mean {r[7]:.2f} max {r[8]:.2f}; "
      f"queued resets/block mean {r[9]:.1f} max {r[10]:.0f}")
