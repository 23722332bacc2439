"""Timeline of one steady-state r6_step launch, per wave and tile (dev tool; needs the R6_TRACE build:
scripts/build_variant.sh trace -DR6_TRACE).  Stamps (100 MHz): a = before the tile's loads are issued, b = loads
arrived, c = compute + stores issued, d = stores acknowledged; HW_ID / XCC_ID place the wave on a SIMD."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

integ = sys.argv[1] if len(sys.argv) > 1 else "rk4"
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "rl_rocket_6dof_amd/csrc/variants/libr6dof_trace.so")
dev = torch.device("cuda:0")
n = 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), device=dev, auto_reset=True, seed=42, integrator=integ,
                        track_episodes=True, record_flags=False, lib_path=lib)
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(2)]
env.reset()
for k in range(1500):
    env.step_async(pool[k % 2])
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
for rep in range(3):
    env.stats.zero_()
    env.step_async(pool[rep % 2])
    torch.cuda.synchronize()
    raw = env.stats.view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1) & 0xFFFFFFFF
    tr = raw[: (raw.size // 24) * 24].reshape(-1, 24)
    tr = tr[tr[:, 3] > 0]
    nt = tr[:, 3]
    hw, xcc, t0w = tr[:, 0], tr[:, 1] & 0xF, tr[:, 2]
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
    t0 = t0w.min()
    us = lambda x: (x - t0) / 100.0
    st = np.stack([us(tr[:, 4 * k: 4 * k + 4]) for k in range(1, 6)], axis=1)      # (waves, tile, 4)
    valid = np.arange(1, 6)[None, :] <= nt[:, None]
    end = np.array([st[i, nt[i] - 1, 3] for i in range(len(tr))])
    np.save(os.path.join(ROOT, "gpurun_out", f"trace_raw_{rep}.npy"), raw[: (raw.size // 24) * 24].reshape(-1, 24)[: 5120 * 2].astype(np.uint32))
    print(f"--- launch {rep}: {len(tr)} waves on {len(np.unique(key))} SIMDs ({len(np.unique(xcc))} XCCs); tiles/wave {np.bincount(nt)[1:]}")
    print(f"kernel entry spread {us(t0w).max():.2f} us; wave end: min {end.min():.2f} p10 {np.percentile(end,10):.2f} median {np.median(end):.2f} p90 {np.percentile(end,90):.2f} max {end.max():.2f}")
    for k in range(4):
        m = valid[:, k]
        if m.any():
            a, b, c, d = (st[m, k, j] for j in range(4))
            print(f"  tile {k+1}: n={m.sum():5d}  start {a.mean():6.2f}  load wait {np.mean(b-a):5.2f} (p90 {np.percentile(b-a,90):5.2f})  compute {np.mean(c-b):5.2f} (p10 {np.percentile(c-b,10):5.2f} p90 {np.percentile(c-b,90):5.2f})  store drain {np.mean(d-c):5.2f}")
    # per SIMD: when is at least one wave in its compute phase; first data arrival; last end
    first_b, last_c, busy, ends = [], [], [], []
    grid = np.arange(0, end.max() + 0.1, 0.05)
    occ = np.zeros_like(grid)
    alive = np.zeros_like(grid)
    for kk in np.unique(key):
        w = np.nonzero(key == kk)[0]
        iv = [(st[i, k, 1], st[i, k, 2]) for i in w for k in range(nt[i])]
        cov = np.zeros_like(grid, dtype=bool)
        for (b_, c_) in iv:
            cov |= (grid >= b_) & (grid < c_)
        occ += cov
        for i in w:
            alive += (grid < end[i])
        first_b.append(min(x[0] for x in iv))
        last_c.append(max(x[1] for x in iv))
        busy.append(cov.sum() * 0.05)
        ends.append(end[w].max())
    first_b, last_c, busy, ends = map(np.array, (first_b, last_c, busy, ends))
    print(f"per SIMD: first data at {first_b.mean():.2f} (max {first_b.max():.2f}); some wave in compute for {busy.mean():.2f} us (min {busy.min():.2f} max {busy.max():.2f}); "
          f"last compute ends {last_c.mean():.2f} (p10 {np.percentile(last_c,10):.2f} p90 {np.percentile(last_c,90):.2f} max {last_c.max():.2f})")
    nS = len(first_b)
    print("fraction of SIMDs with a wave in compute, by time:", " ".join(f"{t:.0f}us:{occ[int(t/0.05)]/nS:.2f}" for t in range(0, int(end.max()) + 1, 2)))
    print("mean waves alive per SIMD, by time:              ", " ".join(f"{t:.0f}us:{alive[int(t/0.05)]/nS:.1f}" for t in range(0, int(end.max()) + 1, 2)))
    wgid = np.nonzero(raw[: (raw.size // 24) * 24].reshape(-1, 24)[:, 3] > 0)[0] // 4      # blockIdx of each traced wave
    print("blockIdx % 8 -> XCC:", " ".join(f"{b}:{np.bincount(xcc[wgid % 8 == b], minlength=8).argmax()}" for b in range(8)),
          "| CU slot order: blockIdx//256 of the workgroups of one CU:", sorted(set((wgid[(xcc == 0) & (se == 0) & (sh == 0) & (cu == 0)] // 256).tolist()))[:8])
    for x in range(8):
        m = xcc == x
        lw = np.concatenate([st[m & valid[:, k], k, 1] - st[m & valid[:, k], k, 0] for k in range(4)])
        cp = np.concatenate([st[m & valid[:, k], k, 2] - st[m & valid[:, k], k, 1] for k in range(4)])
        sd = np.concatenate([st[m & valid[:, k], k, 3] - st[m & valid[:, k], k, 2] for k in range(4)])
        print(f"   XCC {x}: waves {m.sum():4d} tiles {nt[m].sum():5d}  load wait {lw.mean():5.2f}  compute {cp.mean():5.2f}  store drain {sd.mean():5.2f}  first data {st[m, 0, 1].mean():5.2f}  wave end mean {end[m].mean():5.2f} max {end[m].max():5.2f}")
    byx = [ends[(np.unique(key) // (8 * 2 * 64)) == x].mean() if ((np.unique(key) // (8 * 2 * 64)) == x).any() else float('nan') for x in range(8)]
    print("SIMD end by XCC:", " ".join(f"{v:.1f}" for v in byx))
