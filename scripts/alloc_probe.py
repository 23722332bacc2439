#!/usr/bin/env python3
"""Does the PLACEMENT of an env's arrays decide its step time at large N (dev tool, GPU box)?  Round 4 saw two separately allocated
envs of 2^23 envs differ by 11 % in one process (277 vs 309 us per step).  This builds several envs -- torch's own allocations, and
slabs carved at chosen alignments / spacings -- prints where every array sits and times each env on its own."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 23)
ap.add_argument("--envs", type=int, default=4)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--burnin", type=int, default=600)
ap.add_argument("--pool", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda:0")
n = args.n
cfg = default_env_config()
gen = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(args.pool)]
MB = 1 << 20


def describe(env):
    out = []
    for name in ("state", "epconst", "obs", "reward", "done", "final_obs"):
        t = getattr(env, name)
        if t is not None:
            p = t.data_ptr()
            out.append(f"{name}@{p >> 21:#x}+{(p & (2 * MB - 1)) >> 10}K")
    return " ".join(out)


def time_env(env):
    env.reset()
    for k in range(args.burnin):
        env.step_async(pool[k % args.pool])
    torch.cuda.synchronize()
    res = []
    for r in range(args.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for j in range(args.steps):
            env.step_async(pool[j % args.pool])
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / args.steps * 1e3)
    res.sort()
    return res[len(res) // 2]


def rebase(env, spacing_mb, align_mb, skew_kb):
    """Moves the env's arrays into ONE slab: each array starts `align_mb`-aligned (+ k * skew_kb for the k-th array), arrays are
    `spacing_mb` apart beyond their own size."""
    names = [k for k in ("state", "epconst", "_act", "obs", "reward", "done", "final_obs", "flags") if getattr(env, k, None) is not None]
    sizes = {k: getattr(env, k).numel() * getattr(env, k).element_size() for k in names}
    A = align_mb * MB
    total = sum((s + A - 1) // A * A + spacing_mb * MB + A for s in sizes.values()) + 4 * A
    slab = torch.zeros((total,), dtype=torch.uint8, device=dev)
    base = slab.data_ptr()
    off = (-base) % A
    for k_i, k in enumerate(names):
        old = getattr(env, k)
        o = off + k_i * skew_kb * 1024
        view = slab[o:o + sizes[k]].view(old.dtype).view(old.shape)
        view.copy_(old)
        setattr(env, k, view)
        off = (off + sizes[k] + A - 1) // A * A + spacing_mb * MB
    env._slab = slab
    env._fill_buffers(env._act)


kw = dict(device=dev, record_flags=False, record_terms=False, track_episodes=False, seed=42)
for i in range(args.envs):
    filler = torch.empty((3 * MB * (i + 1) + 12345 * i,), dtype=torch.uint8, device=dev)    # perturbs where the next allocations land
    env = BatchedRocket6DOF(n, cfg, **kw)
    print(f"torch env {i}: {time_env(env):7.2f} us/step   {describe(env)}", flush=True)
    del env
    torch.cuda.empty_cache()
import itertools
grid = list(itertools.product((0, 2, 16, 64, 256), (2,), (0, 36, 100))) if not os.environ.get("ALLOC_GRID") else [tuple(int(x) for x in g.split(",")) for g in os.environ["ALLOC_GRID"].split(";")]
for (sp, al, sk) in grid:
    env = BatchedRocket6DOF(n, cfg, **kw)
    rebase(env, sp, al, sk)
    print(f"slab spacing {sp:3d} MB align {al:4d} MB skew {sk:3d} KB: {time_env(env):7.2f} us/step   {describe(env)}", flush=True)
    del env
    torch.cuda.empty_cache()
