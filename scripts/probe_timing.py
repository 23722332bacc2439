import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
cfg = default_env_config(); dev = torch.device("cuda:0")
for n in (1 << 20, 1 << 22):
    for pool_n in (4, 8):
        env = BatchedRocket6DOF(n, cfg, device=dev, record_flags=False, record_terms=False, track_episodes=False, seed=42)
        gen = torch.Generator(device=dev).manual_seed(1)
        pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(pool_n)]
        env.reset()
        for k in range(300): env.step_async(pool[k % pool_n])
        torch.cuda.synchronize()
        x0 = env.state[0, :n].double().sum().item()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); ev0.record()
        for j in range(50): env.step_async(pool[j % pool_n])
        ev1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
        x1 = env.state[0, :n].double().sum().item()
        nd = int(env.done.sum().item())
        print(f"n={n} pool={pool_n}: event {ev0.elapsed_time(ev1)/50*1e3:.1f} us/step, wall {(t1-t0)/50*1e6:.1f} us/step, sum(x) {x0:.6e} -> {x1:.6e}, done last step {nd}", flush=True)
        del env, pool
