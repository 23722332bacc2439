#!/usr/bin/env python3
"""A/B of kernel builds on ONE set of buffers (dev tool, GPU box): a single env whose library handle is swapped between timing
rounds, so that every build steps the same memory (separate envs differ by +-3 % from their allocations alone: DRAM channel
phase of the rows).  Rounds are interleaved and every other round walks the builds backwards; reported per build: median,
lower-half mean, min of the per-round means (HIP events around `--steps` launches).
    python scripts/ab_swap.py --libs a=path,b=path [--n N] [--integrator rk4|euler|rk45] [--wrapped] [--envs 3]
--envs K repeats the whole measurement on K separately allocated envs and prints each, then the pooled figures."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--rounds", type=int, default=21)
ap.add_argument("--burnin", type=int, default=2000)
ap.add_argument("--libs", required=True)
ap.add_argument("--integrator", default="rk4")
ap.add_argument("--dt", type=float, default=None)
ap.add_argument("--wrapped", action="store_true")
ap.add_argument("--envs", type=int, default=2)
ap.add_argument("--pool", type=int, default=2)
ap.add_argument("--reset-on-swap", type=int, default=0, help="> 0: after every swap the env is reset and stepped this many times before timing (builds that "
                                                             "interpret the buffers differently)")
ap.add_argument("--shapes", default="base", help="';'-separated r6_tune settings applied per timing round, e.g. 'base;grid=1024;grid=1024,even=0.5'")
args = ap.parse_args()
cfg = default_env_config()
dev = torch.device("cuda:0")
n = args.n
libs = [x.split("=") for x in args.libs.split(",")]
handles = {name: _lib.load(path) for name, path in libs}
KEYS = {"even": "even_xcd_share", "grid": "grid_blocks", "sweep": "sweep", "flags": "sweep_flags", "tpw": "balanced_tiles_per_wave", "nt": "streaming_stores", "defer": "rk45_defer"}
DEFAULTS = {"even_xcd_share": -1.0, "grid_blocks": 0, "sweep": -1, "sweep_flags": 0, "balanced_tiles_per_wave": 0, "streaming_stores": -1, "rk45_defer": 6}


def apply(shape, lib):
    vals = dict(DEFAULTS)
    if shape != "base":
        for kv in shape.split(","):
            k, v = kv.split("=")
            vals[KEYS[k]] = float(v)
    for k, v in vals.items():
        try:
            _lib.tune(k, v, lib)
        except _lib.R6Error:
            if k != "rk45_defer":      # a library of an earlier round does not know the key
                raise


shapes = args.shapes.split(";")
names = [f"{name}|{sh}" for name, _ in libs for sh in shapes]
gen = torch.Generator(device=dev).manual_seed(1)
pooled = {k: [] for k in names}
keep = []
for ei in range(args.envs):
    keep.append(torch.empty((1 << 20) * (1 + 3 * ei), device=dev))   # shifts the next allocations
    kw = dict(device=dev, record_flags=False, record_terms=False, track_episodes=False, seed=42 + ei, integrator=args.integrator, timestep=args.dt)
    if args.wrapped:
        kw.update(track_episodes=True, remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500)
    env = BatchedRocket6DOF(n, cfg, **kw)
    pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(args.pool)]
    env.reset()
    for k in range(args.burnin):
        env.step_async(pool[k % args.pool])
    torch.cuda.synchronize()
    res = {k: [] for k in names}
    for r in range(args.rounds):
        for name in (names[::-1] if r % 2 else names):
            env.lib = handles[name.split("|")[0]]
            apply(name.split("|")[1], env.lib)
            if args.reset_on_swap > 0:
                env.reset()
                for j in range(args.reset_on_swap):
                    env.step_async(pool[j % args.pool])
            for j in range(20):
                env.step_async(pool[j % args.pool])
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for j in range(args.steps):
                env.step_async(pool[j % args.pool])
            ev1.record()
            torch.cuda.synchronize()
            res[name].append(ev0.elapsed_time(ev1) / args.steps * 1e3)
    for name in names:
        v = sorted(res[name])
        pooled[name] += res[name]
        print(f"env {ei} {name:28s} us/step median {v[len(v) // 2]:7.2f} lower-half mean {sum(v[:(len(v) + 1) // 2]) / ((len(v) + 1) // 2):7.2f} min {v[0]:7.2f}", flush=True)
    del env, pool
    torch.cuda.empty_cache()
b = 201 if args.wrapped else 189
base = None
for name in names:
    v = sorted(pooled[name])
    med = v[len(v) // 2]
    lo = sum(v[:(len(v) + 1) // 2]) / ((len(v) + 1) // 2)
    base = med if base is None else base
    print(f"POOLED {name:28s} us/step median {med:7.2f} ({med - base:+.2f} vs {names[0]}) lower-half mean {lo:7.2f} min {v[0]:7.2f}  frac {b * n / med * 1e-3 / 8000:.3f}", flush=True)
