"""Steps per second of the single-env drop-ins (BASELINE configs[0]: one env, 1000 random-action steps): Rocket6DOF.step() and
Simulator6DOF.step() through their pinned host images, next to the per-tensor read-back they replaced (round 4's form, rebuilt
here from the batched env).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rl_rocket_6dof_amd import BatchedRocket6DOF, Rocket6DOF, Simulator6DOF, default_env_config


def rate_env(env, steps, rng):
    env.reset()
    n, t0 = 0, time.perf_counter()
    while n < steps:
        _, _, done, _ = env.step(rng.uniform(-1, 1, 3).astype(np.float32))
        n += 1
        if done:
            env.reset()
    return n / (time.perf_counter() - t0)


def rate_per_tensor(vec, steps, rng):
    """round 4's Rocket6DOF.step(): obs / reward / done / state / infos each fetched on their own"""
    vec.reset()
    n, t0 = 0, time.perf_counter()
    while n < steps:
        a = rng.uniform(-1, 1, (1, 3)).astype(np.float32)
        obs, rew, done, infos = vec.step(torch.as_tensor(a))
        state = vec.get_state()[0].cpu().numpy().astype(np.float64)
        out = (np.asarray(obs[0].cpu().numpy(), dtype=np.float32), float(rew[0].item()), bool(done[0].item()), infos[0], state)
        n += 1
        if out[2]:
            vec.reset()
    return n / (time.perf_counter() - t0)


def main(steps=1000, repeats=5):
    cfg = default_env_config()
    rng = np.random.RandomState(0)
    res = {"steps": steps, "repeats": repeats}
    for integ in ("rk4", "rk45"):
        env = Rocket6DOF(**cfg, integrator=integ)
        rate_env(env, 200, rng)
        res[f"rocket6dof_{integ}"] = float(np.median([rate_env(env, steps, rng) for _ in range(repeats)]))
    vec = BatchedRocket6DOF(1, cfg, auto_reset=False, record_terms=True, record_flags=True, seed=42)
    rate_per_tensor(vec, 200, rng)
    res["rocket6dof_rk4_per_tensor_readback"] = float(np.median([rate_per_tensor(vec, steps, rng) for _ in range(repeats)]))
    sim = Simulator6DOF([100, 100, 100, 0, 3, 4, 1, 1, 0, 0, 0, 0, 0, 50e3], dt=0.1)

    def rate_sim(k):
        t0 = time.perf_counter()
        for _ in range(k):
            sim.state = np.array([100, 100, 100, 0, 3, 4, 1, 1, 0, 0, 0, 0, 0, 50e3], dtype=np.float64)
            sim.step([0.0, 0.0, 1.0])
        return k / (time.perf_counter() - t0)

    rate_sim(200)
    res["simulator6dof_rk4"] = float(np.median([rate_sim(steps) for _ in range(repeats)]))
    res["reference_python_steps_per_s"] = 730.0
    # SB3's collect_rollouts loop (numpy actions in, numpy obs / rewards / dones out, infos of the finished envs read): calls per second
    for n in (8, 256, 4096, 65536):
        for mirror in (True, False):
            env = BatchedRocket6DOF(n, cfg, numpy_io=True, host_mirror=mirror, remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, seed=1)
            env.reset()
            act = rng.uniform(-1, 1, (n, 3)).astype(np.float32)

            def loop(k):
                t0 = time.perf_counter()
                for _ in range(k):
                    obs, rew, dones, infos = env.step(act)
                    for i in np.nonzero(dones)[0]:
                        infos[int(i)].get("terminal_observation")
                return k / (time.perf_counter() - t0)

            loop(100)
            res[f"numpy_io_{n}_{'mirror' if mirror else 'per_tensor'}_calls_per_s"] = float(np.median([loop(300) for _ in range(repeats)]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
