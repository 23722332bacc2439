#!/usr/bin/env python3
"""Does the length of the untimed pre-roll decide whether the 20 timed launches of the driver's command shape sit in a slow episode?
(dev tool, GPU box; round 5: 5 of 11 driver-shape runs with the 1 s pre-roll read 1.07-1.12 x their own steady median, more than the
15-25 % of 20-launch windows that slow episodes cover in a continuous run.)  Per trial, from an idle chip (2 s of sleep): stepping for P
seconds as bench.py's preroll does (batches of 200 launches, a synchronize between them), 5 warm-up launches, barrier, 20 launches
between two events, then 1000 launches in chunks of 20 for the steady median of the same moment.  P is drawn in random order."""
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
P = [float(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0.25,0.5,1.0,1.5,2.5,4.0".split(","))]
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
order = [p for p in P for _ in range(trials)]
random.Random(7).shuffle(order)
res = {p: [] for p in P}
for p in order:
    time.sleep(idle)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < p:
        run(200); sync()
    sync(); run(5); sync(); sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(20); e1.record(); sync(); sync()
    t20 = e0.elapsed_time(e1) / 20 * 1e3
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(51)]
    ev[0].record()
    for c in range(50):
        run(20); ev[c + 1].record()
    sync()
    ch = sorted(ev[c].elapsed_time(ev[c + 1]) / 20 * 1e3 for c in range(50))
    res[p].append({"t20_us": round(t20, 2), "steady_median_us": round(ch[25], 2), "ratio": round(t20 / ch[25], 3), "steady_p90_us": round(ch[45], 2)})
    print(f"preroll {p:4.2f} s: 20 launches {t20:6.2f} us vs steady median {ch[25]:6.2f} (p90 {ch[45]:6.2f})  ratio {t20 / ch[25]:.3f}", flush=True)
out = {str(p): {"ratios": [r["ratio"] for r in v], "median_ratio": float(np.median([r["ratio"] for r in v])), "share_above_1.06": float(np.mean([r["ratio"] > 1.06 for r in v])),
                "t20_us": [r["t20_us"] for r in v], "steady_median_us": [r["steady_median_us"] for r in v]} for p, v in res.items()}
print(json.dumps(out))
