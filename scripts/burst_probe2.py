#!/usr/bin/env python3
"""Slow 20-launch chunks, part 2: do they replay with the data, and what do the chip's sensors say meanwhile?  (dev tool, GPU box)

Two runs of the SAME deterministic sequence (same seed, same actions, same number of launches before the series): chunk means of
run A against run B -- bursts at the same chunk indices are properties of the data, bursts at other indices are the chip's.
During each series a side thread samples whatever sensors sysfs lets an ordinary user read (hwmon power / sclk / mclk, gpu_busy_percent)
with host timestamps; the series' host start time and the chunks' cumulative GPU times place the samples on the chunk axis."""
import argparse, glob, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--chunks", type=int, default=300)
ap.add_argument("--chunk", type=int, default=20)
ap.add_argument("--pre", type=int, default=20000, help="launches before the series (fixed count: the sequence must replay)")
args = ap.parse_args()
dev = torch.device("cuda:0")
n = args.n


def sensor_files():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for pat, key in (("hwmon/hwmon*/power1_average", "power_uW"), ("hwmon/hwmon*/power1_input", "power_in_uW"), ("hwmon/hwmon*/freq1_input", "sclk_Hz"),
                         ("hwmon/hwmon*/freq2_input", "mclk_Hz"), ("gpu_busy_percent", "busy"), ("hwmon/hwmon*/temp1_input", "temp_mC")):
            for f in glob.glob(os.path.join(card, pat)):
                try:
                    open(f).read()
                    out[f"{os.path.basename(os.path.dirname(card))}:{key}"] = f
                except OSError:
                    pass
    return out


class Sampler(threading.Thread):
    def __init__(self, files):
        super().__init__(daemon=True)
        self.files, self.rows, self.stop = files, [], False

    def run(self):
        while not self.stop:
            row = {"t": time.perf_counter()}
            for k, f in self.files.items():
                try:
                    row[k] = float(open(f).read().split()[0])
                except (OSError, ValueError, IndexError):
                    pass
            self.rows.append(row)
            time.sleep(0.0005)


files = sensor_files()
print("sensors:", list(files), file=sys.stderr, flush=True)


def one_run(tag):
    env = BatchedRocket6DOF(n, default_env_config(), device=dev, auto_reset=True, record_flags=True, record_terms=False, track_episodes=False, seed=42)
    gen = torch.Generator(device=dev).manual_seed(1234)
    pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(2)]
    env.reset()
    for j in range(args.pre):
        env.step_async(pool[j % 2])
        if j % 500 == 499:
            torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.chunks + 1)]
    ground = torch.zeros(args.chunks, device=dev)
    bounds = torch.zeros(args.chunks, device=dev)
    smp = Sampler(files)
    torch.cuda.synchronize()
    smp.start()
    t_start = time.perf_counter()
    ev[0].record()
    for c in range(args.chunks):
        for j in range(args.chunk):
            env.step_async(pool[j % 2])
        ground[c] = (env.flags & 1).sum()
        bounds[c] = ((env.flags >> 1) & 1).sum()
        ev[c + 1].record()
    torch.cuda.synchronize()
    smp.stop = True
    smp.join()
    us = np.array([ev[c].elapsed_time(ev[c + 1]) / args.chunk * 1e3 for c in range(args.chunks)])
    edges = t_start + np.concatenate([[0.0], np.cumsum(us * args.chunk * 1e-6)])
    sens = {}
    for k in files:
        acc = [[] for _ in range(args.chunks)]
        for r in smp.rows:
            if k in r:
                c = int(np.searchsorted(edges, r["t"]) - 1)
                if 0 <= c < args.chunks:
                    acc[c].append(r[k])
        sens[k] = [float(np.mean(a)) if a else None for a in acc]
    del env
    torch.cuda.empty_cache()
    return {"us": [round(float(x), 2) for x in us], "ground_last_step": ground.cpu().tolist(), "bounds_last_step": bounds.cpu().tolist(),
            "sensors": sens, "samples": len(smp.rows)}


A = one_run("A")
B = one_run("B")
ua, ub = np.array(A["us"][1:]), np.array(B["us"][1:])
out = {"n": n, "chunk": args.chunk, "pre": args.pre, "A": A, "B": B, "corr_A_B": float(np.corrcoef(ua, ub)[0, 1]),
       "identical_event_counts": A["ground_last_step"] == B["ground_last_step"]}
print("corr(A, B) of the chunk means:", out["corr_A_B"], " event counts replay:", out["identical_event_counts"], file=sys.stderr)
slow_a = set(np.nonzero(ua > np.median(ua) * 1.06)[0].tolist())
slow_b = set(np.nonzero(ub > np.median(ub) * 1.06)[0].tolist())
print(f"slow chunks (> 1.06 x median): A {len(slow_a)}, B {len(slow_b)}, common {len(slow_a & slow_b)}", file=sys.stderr)
for k, v in A["sensors"].items():
    vals = [(x, u) for x, u in zip(v[1:], ua) if x is not None]
    if len(vals) > 20:
        xs, us_ = np.array([a for a, _ in vals]), np.array([b for _, b in vals])
        if xs.std() > 0:
            print(f"  {k}: mean {xs.mean():.4g}  min {xs.min():.4g}  max {xs.max():.4g}  corr with chunk time {np.corrcoef(xs, us_)[0, 1]:+.2f}", file=sys.stderr)
        else:
            print(f"  {k}: constant {xs.mean():.4g}", file=sys.stderr)
print(json.dumps(out))
