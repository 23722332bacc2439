"""Where is fixed-step RK4 (dt = 0.1, the kernel's default) still a twin of the reference's adaptive RK45 (rtol 1e-3)?
Worst scaled one-step difference over random states, binned by |omega| (the only fast time scale of the model)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as onp
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

cfg = default_env_config()
P = onp.OracleParams.from_config(cfg)
rng = np.random.default_rng(3)
n = 1200
s = np.zeros((n, 14))
s[:, 0] = rng.uniform(50.0, 2500.0, n)
s[:, 1:3] = rng.uniform(-2000.0, 2000.0, (n, 2))
s[:, 3:6] = rng.uniform(-200.0, 200.0, (n, 3))
q = rng.normal(size=(n, 4))
s[:, 6:10] = q / np.linalg.norm(q, axis=1, keepdims=True)
wmag = np.repeat([0.05, 0.2, 0.5, 1.0, 2.0, 4.0], n // 6)
w = rng.normal(size=(n, 3))
s[:, 10:13] = w / np.linalg.norm(w, axis=1, keepdims=True) * wmag[:, None]
s[:, 13] = rng.uniform(25e3, 45e3, n)
s = s.astype(np.float32)
a = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
out = {}
for integ in ("rk4", "rk45"):
    env = BatchedRocket6DOF(n, cfg, integrator=integ, auto_reset=False)
    env.set_state(s)
    env.step(torch.as_tensor(a))
    out[integ] = env.get_state().cpu().numpy().astype(np.float64)
ref = np.array([onp.sim_step_adaptive(s[i].astype(np.float64), onp.denormalize_action(a[i]), float(s[i, 13]), 0.1)[0] for i in range(n)])
tight = []
from scipy.integrate import solve_ivp
for i in range(n):
    u = onp.denormalize_action(a[i])
    sol = solve_ivp(lambda t, y: onp.rhs(y, u, float(s[i, 13])), (0.0, 0.1), s[i].astype(np.float64), rtol=1e-12, atol=1e-12)
    y = sol.y[:, -1].copy(); y[6:10] /= np.linalg.norm(y[6:10]); tight.append(y)
tight = np.array(tight)
sc = lambda x, y: np.max(np.abs(x - y) / np.maximum(np.abs(y), P.normalizer), axis=1)
print("|omega| rad/s : worst scaled one-step difference")
for wm in (0.05, 0.2, 0.5, 1.0, 2.0, 4.0):
    m = wmag == wm
    print(f"  {wm:4.2f}: rk4 kernel vs reference RK45(rtol 1e-3) {sc(out['rk4'][m], ref[m]).max():.1e} | rk45 kernel vs reference {sc(out['rk45'][m], ref[m]).max():.1e}"
          f" | reference vs exact {sc(ref[m], tight[m]).max():.1e} | rk4 kernel vs exact {sc(out['rk4'][m], tight[m]).max():.1e}")
