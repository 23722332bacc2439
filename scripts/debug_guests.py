#!/usr/bin/env python3
"""Which envs differ between rk45_defer 0 and 6 (dev tool, GPU box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config, _lib
n, grid = 150037, 48
cfg = default_env_config()
gen = torch.Generator(device="cuda:0").manual_seed(7)
acts = [(torch.rand((n, 3), generator=gen, device="cuda:0") * 2 - 1) for _ in range(3)]
res = {}
_lib.tune("grid_blocks", grid)
for defer in (0, 6):
    _lib.tune("rk45_defer", defer)
    env = BatchedRocket6DOF(n, cfg, integrator="rk45", seed=3, record_flags=True, record_terms=True, record_nfev=True, track_episodes=True)
    env.reset()
    out = []
    for k in range(3):
        env.step_async(acts[k])
        torch.cuda.synchronize()
        out.append({k2: getattr(env, k2).clone().cpu().numpy() for k2 in ("obs", "reward", "done", "flags", "nfev", "terms", "state", "ep_step", "ep_return")})
    res[defer] = out
    del env
_lib.tune("rk45_defer", 6); _lib.tune("grid_blocks", 0)
for k in range(3):
    a, b = res[0][k], res[6][k]
    for name in a:
        x, y = a[name], b[name]
        if name in ("terms", "state"):
            x, y = x[:, :n].T, y[:, :n].T
        neq = np.nonzero((x != y).reshape(n, -1).any(1))[0]
        if len(neq):
            print(f"step {k} {name}: {len(neq)} envs differ, first {neq[:12]}, lane {neq[:12] % 64}, tile {neq[:12] // 64}")
            if name in ("reward", "terms"):
                for i in neq[:6]:
                    print("   env", i, "nfev", a["nfev"][i], b["nfev"][i], "reward", a["reward"][i], b["reward"][i], "terms0", a["terms"][:, i], "terms6", b["terms"][:, i], "done", a["done"][i], b["done"][i])
    print("step", k, "envs with nfev >= 20:", int((a["nfev"] >= 20).sum()))
