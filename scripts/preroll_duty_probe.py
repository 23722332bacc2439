#!/usr/bin/env python3
"""Does the DUTY CYCLE of the pre-roll decide whether the 20 timed launches sit in a slow episode?  (dev tool, GPU box)  Slow episodes are
clock dips under sustained full load and do not occur when launches are spaced out.  Per trial, from an idle chip:
1 s of stepping in mode M, 5 warm-up launches, barrier, 20 launches between two events, then 1000 launches for the steady median.
  full     batches of 200 launches, a synchronize between them (bench.py's pre-roll)
  sync1    a synchronize after every launch
  sync4    a synchronize after every 4 launches
  gap      a synchronize and 30 us of host sleep after every launch"""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
dev = torch.device("cuda:0"); n = 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), device=dev, record_flags=False, record_terms=False, track_episodes=False, seed=42)
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)]
env.reset()
sync = lambda: torch.cuda.synchronize(dev)
def run(k):
    for j in range(k):
        env.step_async(pool[j % 2])
run(2000); sync()
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["full", "sync1", "sync4", "gap"]
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
order = [m for m in modes for _ in range(trials)]
random.Random(11).shuffle(order)
res = {m: [] for m in modes}
for m in order:
    time.sleep(2.0)
    t0 = time.perf_counter(); launched = 0
    while time.perf_counter() - t0 < 1.0:
        if m == "full":
            run(200); sync(); launched += 200
        elif m == "sync1":
            run(1); sync(); launched += 1
        elif m == "sync4":
            run(4); sync(); launched += 4
        else:
            run(1); sync(); launched += 1
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < 30e-6:
                pass
    sync(); run(5); sync(); sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(20); e1.record(); sync(); sync()
    t20 = e0.elapsed_time(e1) / 20 * 1e3
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(51)]
    ev[0].record()
    for c in range(50):
        run(20); ev[c + 1].record()
    sync()
    ch = [ev[c].elapsed_time(ev[c + 1]) / 20 * 1e3 for c in range(50)]
    med = sorted(ch)[25]
    res[m].append({"t20_us": round(t20, 2), "steady_median_us": round(med, 2), "ratio": round(t20 / med, 3), "launches_in_preroll": launched,
                   "first_chunks_behind_it_us": [round(x, 1) for x in ch[:6]]})
    print(f"{m:6s} ({launched:5d} launches in 1 s): 20 launches {t20:6.2f} us vs steady median {med:6.2f}  ratio {t20 / med:.3f}   next chunks {[round(x, 1) for x in ch[:6]]}", flush=True)
out = {m: {"t20_us": [r["t20_us"] for r in v], "median_t20_us": float(np.median([r["t20_us"] for r in v])), "max_t20_us": float(np.max([r["t20_us"] for r in v])),
           "share_above_1.06_x_steady_median": float(np.mean([r["ratio"] > 1.06 for r in v])), "launches_in_preroll": int(np.median([r["launches_in_preroll"] for r in v]))} for m, v in res.items()}
print(json.dumps(out))
