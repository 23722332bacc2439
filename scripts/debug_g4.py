"""Debug helper (GPU box): where does the teacher-forced error of the ground-event fixture come from?"""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from oracle import oracle_np as onp
from rl_rocket_6dof_amd import BatchedRocket6DOF

for fx in ("g4_ground_events.npz", "g3_env_rollouts.npz"):
    G = load_golden(fx)
    cfg = json.loads(str(G["config"]))
    P = onp.OracleParams.from_config(cfg)
    n = len(G["reward"])
    env = BatchedRocket6DOF(n, cfg, auto_reset=False, record_terms=True)
    env.set_state(G["pre"].astype(np.float32), m0=G["m0"].astype(np.float32))
    obs, rew, done, _ = env.step(torch.as_tensor(G["action"], device=env.device))
    post = env.get_state().cpu().numpy().astype(np.float64)
    err = np.abs(post - G["post"]) / np.maximum(np.abs(G["post"]), P.normalizer[None, :])
    print(fx, "per-component worst:", np.array2string(err.max(0), precision=2))
    ev = G["status"] == 1
    print("  events only:", np.array2string(err[ev].max(0), precision=2) if ev.any() else None)
    print("  non-events :", np.array2string(err[~ev].max(0), precision=2))
    order = np.argsort(-err.max(1))[:6]
    for i in order:
        j = err[i].argmax()
        o = onp.env_step(G["pre"][i].astype(np.float32).astype(np.float64), float(np.float32(G["m0"][i])), G["action"][i], P, "rk4")
        print(f"  env {i} comp {j} status {G['status'][i]} gpu {post[i,j]:.9g} ref {G['post'][i,j]:.9g} f64-oracle(f32 inputs) {o['post'][j]:.9g} "
              f"pre x {G['pre'][i,0]:.4g} vx {G['pre'][i,3]:.4g} gpu x {post[i,0]:.3g} ref x {G['post'][i,0]:.3g}")
    terms = env.terms[:, :n].t().cpu().numpy()
    te = np.abs(terms - G["terms"])
    print("  reward-term abs err worst:", np.array2string(te.max(0), precision=2))
