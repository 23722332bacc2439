#!/usr/bin/env python3
"""Where do the slow 20-launch chunks of the headline workload come from -- the data or the chip?  (dev tool, GPU box; round 5)

Three series of 20-launch chunk means (events between chunks, one stream, back to back), each after >= 1 s of stepping:
  step        the bench's env (2^20 envs, RK4, auto-reset): chunk mean us, share of envs that finished in the chunk's last step
  frozen      the same kernel on a state restored before every launch is NOT possible in place; instead auto_reset = False on a
              population of long flights: no resets, (almost) constant work per launch
  membench    scripts/membench.so's k_pattern: the step's memory accesses without arithmetic: constant work
Output: JSON on stdout."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--chunks", type=int, default=300)
ap.add_argument("--chunk", type=int, default=20)
ap.add_argument("--preroll", type=float, default=1.0)
args = ap.parse_args()
dev = torch.device("cuda:0")
n = args.n
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(2)]


def series(launch, after_chunk=None):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.preroll:
        for j in range(200):
            launch(j)
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.chunks + 1)]
    extra = []
    torch.cuda.synchronize()
    ev[0].record()
    for c in range(args.chunks):
        for j in range(args.chunk):
            launch(j)
        if after_chunk is not None:
            extra.append(after_chunk())
        ev[c + 1].record()
    torch.cuda.synchronize()
    us = [ev[c].elapsed_time(ev[c + 1]) / args.chunk * 1e3 for c in range(args.chunks)]
    return us, [float(x) for x in extra]


out = {"n": n, "chunk": args.chunk}
env = BatchedRocket6DOF(n, default_env_config(), device=dev, auto_reset=True, record_flags=False, record_terms=False, track_episodes=False, seed=42)
env.reset()
for j in range(2000):
    env.step_async(pool[j % 2])
us, done = series(lambda j: env.step_async(pool[j % 2]), lambda: env.done.sum() / n)
out["step"] = {"us": [round(x, 2) for x in us], "done_share": [round(x, 5) for x in done]}
s = sorted(us)
print("step: median", s[len(s) // 2], "p10", s[len(s) // 10], "p90", s[9 * len(s) // 10], file=sys.stderr, flush=True)
import numpy as np
print("step: corr(us, done share)", float(np.corrcoef(us, done)[0, 1]), file=sys.stderr, flush=True)

# constant work: no auto-reset, hovering population (state rewritten each chunk from a saved copy so nobody ever lands)
env2 = BatchedRocket6DOF(n, default_env_config(), device=dev, auto_reset=False, record_flags=False, record_terms=False, track_episodes=False, seed=43)
env2.reset()
saved = env2.state.clone()
def restore():
    env2.state.copy_(saved)
    return 0.0
us2, _ = series(lambda j: env2.step_async(pool[j % 2]), restore)
out["frozen_population_no_reset"] = {"us": [round(x, 2) for x in us2], "note": "chunk time includes one 60 MB state copy per 20 launches"}
s = sorted(us2)
print("frozen: median", s[len(s) // 2], "p10", s[len(s) // 10], "p90", s[9 * len(s) // 10], file=sys.stderr, flush=True)

here = os.path.dirname(os.path.abspath(__file__))
try:
    lib = C.CDLL(os.path.join(here, "membench.so"))
    vp = lambda t: C.c_void_p(t.data_ptr())
    st = torch.rand((14, n), device=dev); ep = torch.rand((2, n), device=dev)
    obs = torch.empty((n, 14), device=dev); rew = torch.empty(n, device=dev); dn = torch.empty(n, dtype=torch.uint8, device=dev)
    sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    us3, _ = series(lambda j: lib.mb_pattern(vp(st), vp(st), vp(ep), vp(pool[j % 2]), vp(obs), vp(rew), vp(dn), C.c_int64(n), C.c_int64(n), 1280, sh, 0))
    out["membench_pattern"] = {"us": [round(x, 2) for x in us3]}
    s = sorted(us3)
    print("membench: median", s[len(s) // 2], "p10", s[len(s) // 10], "p90", s[9 * len(s) // 10], file=sys.stderr, flush=True)
except OSError as e:
    out["membench_pattern"] = {"error": str(e)}
print(json.dumps(out))
