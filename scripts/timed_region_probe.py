#!/usr/bin/env python3
"""Is a 20-launch timed region behind the bench's sequence of synchronizes more exposed to clock dips than a chunk of a continuous run?
(dev tool, GPU box)  Interleaved repetitions of
  A  bench.py as it was: [200 launches, sync] x 3, then barrier (2 syncs), 5 warm-up launches, barrier, barrier, 20 timed launches, barrier
  B  the same launches with no synchronize between the pre-roll and the warm-up: [600 launches + 5 warm-up], ONE barrier, 20 timed, barrier
  C  continuous: 600 launches, then 20 launches between two events, no synchronize before them
HIP-event time of the 20 launches, us per launch."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
dev = torch.device("cuda:0"); n = 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), device=dev, record_flags=False, record_terms=False, track_episodes=False, seed=42)
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)]
env.reset()
for k in range(30000):
    env.step_async(pool[k % 2])
sync = lambda: torch.cuda.synchronize(dev)
def barrier():
    sync(); sync()
def run(k):
    for j in range(k):
        env.step_async(pool[j % 2])
def timed20(pre_sync):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if pre_sync:
        barrier()
    e0.record(); run(20); e1.record()
    barrier()
    return e0.elapsed_time(e1) / 20 * 1e3
res = {"A": [], "B": [], "C": []}
for rep in range(40):
    for mode in ("A", "B", "C") if rep % 2 == 0 else ("C", "B", "A"):
        if mode == "A":
            for _ in range(3):
                run(200); sync()
            barrier(); run(5); barrier()
            res["A"].append(timed20(True))
        elif mode == "B":
            run(605)
            res["B"].append(timed20(True))
        else:
            run(605)
            res["C"].append(timed20(False))
out = {}
for k, v in res.items():
    v = np.array(v)
    out[k] = {"median": float(np.median(v)), "mean": float(v.mean()), "p10": float(np.percentile(v, 10)), "p90": float(np.percentile(v, 90)), "max": float(v.max()),
              "share_above_1.06_x_median_of_C": float((v > 1.06 * np.median(res["C"])).mean()), "values": [round(float(x), 2) for x in v]}
print(json.dumps(out))
