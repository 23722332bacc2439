#!/usr/bin/env python3
"""Shader clock the chip holds while the step kernel runs (dev tool, GPU box; -DR6_CLOCK_PROBE build): per workgroup,
s_memtime cycles over s_memrealtime (100 MHz) ticks between kernel entry and exit, in the steady state of back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
dev = torch.device("cuda:0")
n = 1 << 20
lib = sys.argv[1] if len(sys.argv) > 1 else "rl_rocket_6dof_amd/csrc/variants/libr6dof_clock.so"
for integ in ("rk4", "euler"):
    env = BatchedRocket6DOF(n, default_env_config(), device=dev, integrator=integ, record_flags=False, record_terms=False, track_episodes=True, seed=42, lib_path=lib)
    gen = torch.Generator(device=dev).manual_seed(1)
    pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)]
    env.reset()
    for k in range(3000):
        env.step_async(pool[k % 2])
    for rep in range(4):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for k in range(500):
            env.step_async(pool[k % 2])
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) / 500 * 1e3
        st = env.stats.cpu().numpy()[:1280]
        cyc, ticks = st[:, 0], st[:, 1]
        ghz = cyc / (ticks * 10.0)
        span = (st[:, 3].max() - st[:, 2].min()) / 100.0
        # algorithmic bytes per shader clock and CU: 201 B per env-step here (episode tracking on) over launch-to-launch time
        bpc = 201.0 * n / (us * 1e-6) / 256 / (np.median(ghz) * 1e9)
        print(f"{integ}: clock median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}); workgroup life mean {ticks.mean() / 100:.2f} us, "
              f"launch span {span:.2f} us, launch to launch {us:.2f} us = {201.0 * n / us * 1e-6:.2f} TB/s = {bpc:.2f} B per clock and CU")
