#!/usr/bin/env python3
"""What does a 20-launch timed region see?  (review round 4, Next 1; dev tool, GPU box)

The driver times `bench.py --gpus 1 --steps 20 --warmup 5`: 5 + 20 launches, each block bracketed by synchronize().  This probe
builds the bench's env, then for several (idle gap before the block, amount of stepping before that) combinations records

  * the bracket time of a 20-launch block exactly as bench.timed_loop takes it (two events), and
  * the same block with one event per launch (per-launch durations: where in the block do the slow launches sit?),

followed by 2000 back-to-back launches in 20-launch chunks (the steady distribution a 20-launch sample is drawn from).
Output: one JSON object on stdout (and progress on stderr)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--burnin", type=int, default=2000)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--integrator", default="rk4")
args = ap.parse_args()

dev = torch.device("cuda:0")
t_first = time.perf_counter()
torch.cuda.init()
env = BatchedRocket6DOF(args.n, default_env_config(), device=dev, integrator=args.integrator, auto_reset=True, record_flags=False,
                        record_terms=False, track_episodes=False, seed=42)
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((args.n, 3), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(2)]
env.reset()
for j in range(args.burnin):
    env.step_async(pool[j % 2])
stream = torch.cuda.current_stream(dev)
sync = lambda: torch.cuda.synchronize(dev)


def bracket(steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync()
    t0 = time.perf_counter()
    e0.record(stream)
    for j in range(steps):
        env.step_async(pool[j % 2])
    e1.record(stream)
    sync()
    return e0.elapsed_time(e1) / steps * 1e3, (time.perf_counter() - t0) / steps * 1e6


def per_launch(steps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    sync()
    ev[0].record(stream)
    for j in range(steps):
        env.step_async(pool[j % 2])
        ev[j + 1].record(stream)
    sync()
    return [ev[j].elapsed_time(ev[j + 1]) * 1e3 for j in range(steps)]


def preroll(seconds):
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        for j in range(200):
            env.step_async(pool[j % 2])
        k += 200
        sync()
    return k


out = {"n": args.n, "integrator": args.integrator, "ms_since_first_gpu_touch_at_first_block": None, "blocks": []}
sync()
out["ms_since_first_gpu_touch_at_first_block"] = (time.perf_counter() - t_first) * 1e3
# 1. exactly the driver's shape, right after the burn-in (what BENCH_r04 timed)
w = bracket(5)
b = bracket(20)
out["driver_shape_first"] = {"warmup5_event_us": w[0], "steps20_event_us": b[0], "steps20_wall_us": b[1]}
print("first", out["driver_shape_first"], file=sys.stderr, flush=True)
# 2. idle gap / preroll matrix
for pre_s in (0.0, 1.0, 3.0):
    if pre_s:
        preroll(pre_s)
    for gap in (0.0, 0.002, 0.02, 0.2, 1.0):
        rows = []
        for r in range(args.reps):
            sync()
            if gap:
                time.sleep(gap)
            b = bracket(20)
            sync()
            if gap:
                time.sleep(gap)
            pl = per_launch(20)
            rows.append({"bracket_event_us": b[0], "bracket_wall_us": b[1], "per_launch_us": [round(x, 2) for x in pl]})
        ev = sorted(r["bracket_event_us"] for r in rows)
        firsts = [r["per_launch_us"][0] for r in rows]
        rest = [sum(r["per_launch_us"][5:]) / 15 for r in rows]
        rec = {"preroll_s": pre_s, "idle_gap_s": gap, "bracket_event_us_sorted": [round(x, 2) for x in ev],
               "first_launch_us": [round(x, 1) for x in firsts], "launches_6_to_20_mean_us": [round(x, 2) for x in rest], "rows": rows}
        out["blocks"].append(rec)
        print(f"preroll {pre_s} gap {gap}: bracket {ev[0]:.2f}..{ev[-1]:.2f}  first launch {min(firsts):.1f}..{max(firsts):.1f}  rest {min(rest):.2f}..{max(rest):.2f}",
              file=sys.stderr, flush=True)
# 3. steady distribution of 20-launch chunk means
chunks = 100
ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
sync()
ev[0].record(stream)
for c in range(chunks):
    for j in range(20):
        env.step_async(pool[j % 2])
    ev[c + 1].record(stream)
sync()
d = sorted(ev[c].elapsed_time(ev[c + 1]) / 20 * 1e3 for c in range(chunks))
out["steady_20_launch_chunk_means_us"] = {"min": d[0], "p10": d[10], "median": d[50], "p90": d[90], "max": d[-1]}
print("steady chunks", out["steady_20_launch_chunk_means_us"], file=sys.stderr, flush=True)
print(json.dumps(out))
