#!/usr/bin/env python3
"""Launch-shape A/B of r6_step in ONE process (dev tool, GPU box): the knobs of r6_tune() -- tail pool, fork/join split,
XCD share -- and kernel builds (--libs name=path,...) timed interleaved on the same envs; checks first that every shape
leaves bit-identical state / outputs."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--burnin", type=int, default=2000)
ap.add_argument("--libs", default="")
ap.add_argument("--integrator", default="rk4")
ap.add_argument("--wrapped", action="store_true")
ap.add_argument("--shapes", default="base")
ap.add_argument("--nocheck", action="store_true")
ap.add_argument("--alternate", action="store_true", help="every other round walks the builds / shapes in reverse order (the first one of a round reads slow)")
ap.add_argument("--graph", type=int, default=0, help="> 0: time replays of a HIP graph of this many captured steps instead of eager launches")
args = ap.parse_args()
cfg = default_env_config()
dev = torch.device("cuda:0")
n = args.n
libs = dict(x.split("=") for x in args.libs.split(",")) if args.libs else {"": None}
KEYS = {"even": "even_xcd_share", "grid": "grid_blocks", "sweep": "sweep", "flags": "sweep_flags", "nt": "streaming_stores", "tpw": "balanced_tiles_per_wave"}
DEFAULTS = {"even_xcd_share": -1.0, "grid_blocks": 0, "sweep": -1, "sweep_flags": 0, "streaming_stores": -1, "balanced_tiles_per_wave": 0}


def apply(shape, lib):
    vals = dict(DEFAULTS)
    if shape != "base":
        for kv in shape.split(","):
            k, v = kv.split("=")
            vals[KEYS[k]] = float(v)
    for k, v in vals.items():
        _lib.tune(k, v, lib)


def make(libname, seed=42):
    kw = dict(device=dev, record_flags=False, record_terms=False, track_episodes=False, seed=seed, lib_path=libs.get(libname), integrator=args.integrator)
    if args.wrapped:
        kw.update(track_episodes=True, remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500)
    return BatchedRocket6DOF(n, cfg, **kw)


shapes = args.shapes.split(";")
gen = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)]

if not args.nocheck:
    # bit-exactness: every (build, shape) against the first one, 300 steps from reset (covers the first landing wave when wrapped / the resets)
    ref = None
    for l in libs:
        for sh in shapes:
            e = make(l)
            apply(sh, e.lib)
            e.reset()
            for k in range(300):
                e.step_async(pool[k % 2])
            torch.cuda.synchronize()
            snap = (e.state.clone(), e.obs.clone(), e.reward.clone(), e.done.clone(), e.epconst.clone())
            if e.stats is not None:
                snap += (e.stats.sum(0).clone(),)
            if ref is None:
                ref = snap
            else:
                ok = all(torch.equal(a, b) for a, b in zip(ref[:5], snap[:5]))
                st = "" if len(snap) == 5 else f" stats equal: {bool(torch.equal(ref[5], snap[5]))}"
                print(f"bit-exact {l or 'intree':10s} {sh:32s}: {ok}{st}", flush=True)
                assert ok, (l, sh)
                assert len(snap) == 5 or torch.equal(ref[5], snap[5]), (l, sh, "stats")
            apply("base", e.lib)
            del e
    torch.cuda.empty_cache()

envs = {l: make(l) for l in libs}
for l, e in envs.items():
    apply("base", e.lib)
    e.reset()
    for k in range(args.burnin):
        e.step_async(pool[k % 2])
torch.cuda.synchronize()
import time
names = [(l, sh) for l in libs for sh in shapes]
res = {k: [] for k in names}
host = {k: [] for k in names}
graphs = {}
if args.graph > 0:
    side = torch.cuda.Stream(device=dev)
    for (l, sh) in names:
        e = envs[l]
        apply(sh, e.lib)
        with torch.cuda.stream(side):
            for j in range(4):
                e.step_async(pool[j % 2])     # creates the fork/join resources of this stream outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for j in range(args.graph):
                e.step_async(pool[j % 2])
        graphs[(l, sh)] = g
    torch.cuda.synchronize()
for r in range(args.rounds):
    for (l, sh) in (names[::-1] if (args.alternate and r % 2) else names):
        e = envs[l]
        apply(sh, e.lib)
        for j in range(10):
            e.step_async(pool[j % 2])
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        if args.graph > 0:
            reps = max(1, args.steps // args.graph)
            for j in range(reps):
                graphs[(l, sh)].replay()
            nst = reps * args.graph
        else:
            for j in range(args.steps):
                e.step_async(pool[j % 2])
            nst = args.steps
        ev1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        res[(l, sh)].append(ev0.elapsed_time(ev1) / nst * 1e3)
        host[(l, sh)].append((t1 - t0) / nst * 1e6)
b = 201 if args.wrapped else 189
for (l, sh) in names:
    v = sorted(res[(l, sh)])
    med = v[len(v) // 2]
    lo = sum(v[: (len(v) + 1) // 2]) / ((len(v) + 1) // 2)
    print("   raw:", " ".join(f"{x:.1f}" for x in res[(l, sh)]))
    print(f"{l or 'intree':10s} {sh:34s} us/step median {med:7.2f} lower-half mean {lo:7.2f} min {v[0]:7.2f} max {v[-1]:7.2f}  frac {b * n / med * 1e-3 / 8000:.3f}  host {sorted(host[(l, sh)])[len(v) // 2]:6.1f} us/step", flush=True)
