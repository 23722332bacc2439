#!/usr/bin/env python3
"""Do the slow 20-launch chunks coincide with a lower shader clock?  (dev tool, GPU box; -DR6_CLOCK_PROBE build: every workgroup writes the
s_memtime cycles and the 100 MHz ticks of its life into its row of `stats`)  Back-to-back launches of the bench workload in chunks of 20
with HIP events between the chunks; after every launch the clock rows are copied aside (one small device-to-device copy).  Output: JSON."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
lib = sys.argv[1] if len(sys.argv) > 1 else "rl_rocket_6dof_amd/csrc/variants/libr6dof_clock.so"
dev = torch.device("cuda:0"); n = 1 << 20; chunks, chunk = 200, 20
env = BatchedRocket6DOF(n, default_env_config(), device=dev, record_flags=False, record_terms=False, track_episodes=True, seed=42, lib_path=lib)
gen = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1) for _ in range(2)]
env.reset()
for k in range(30000):          # ~1 s
    env.step_async(pool[k % 2])
RING = 10            # -DR6_CLOCK_RING=10: rows [(step % 10) * 1280 + workgroup]
clk = torch.zeros((chunks * chunk // RING, RING * 1280, 2), dtype=torch.float64, device=dev)
assert env._step_count % RING == 0
ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
torch.cuda.synchronize()
ev[0].record()
i = 0
for c in range(chunks):
    for j in range(chunk):
        env.step_async(pool[j % 2])
        i += 1
        if i % RING == 0:      # one small copy per RING launches (every launch: the copies space the launches out and the slow episodes vanish)
            clk[i // RING - 1].copy_(env.stats[:RING * 1280, :2])
    ev[c + 1].record()
torch.cuda.synchronize()
us = np.array([ev[c].elapsed_time(ev[c + 1]) / chunk * 1e3 for c in range(chunks)])
first = float(us[0])       # the first chunk carries the set-up of the copy path (milliseconds): left out of every statistic below
c = clk.cpu().numpy().reshape(chunks * chunk // RING, RING, 1280, 2)
# ring slot r of copy q holds launch with step % RING == r: the launches of a copy in step order = slots (first_step % RING ...) -- the loop starts at a multiple of RING + 1
c = np.roll(c, -1, axis=1).reshape(chunks * chunk, 1280, 2)
ghz = np.where(c[:, :, 1] > 0, c[:, :, 0] / np.maximum(c[:, :, 1], 1) / 10.0, np.nan)     # cycles / (ticks x 10 ns)
per_launch = np.nanmedian(ghz, axis=1)
per_chunk = per_launch.reshape(chunks, chunk).mean(1)[1:]
life = (np.nanmean(c[:, :, 1], axis=1).reshape(chunks, chunk).mean(1) / 100.0)[1:]           # mean workgroup life, us
us = us[1:]
med = np.median(us)
slow = us > 1.06 * med
out = {"chunks": chunks - 1, "first_chunk_us_left_out": first, "chunk_us": {"median": float(med), "p10": float(np.percentile(us, 10)), "p90": float(np.percentile(us, 90))},
       "clock_ghz_per_chunk": {"median": float(np.median(per_chunk)), "p10": float(np.percentile(per_chunk, 10)), "p90": float(np.percentile(per_chunk, 90)), "min": float(per_chunk.min()), "max": float(per_chunk.max())},
       "corr(chunk time, chunk clock)": float(np.corrcoef(us, per_chunk)[0, 1]), "corr(chunk time, mean workgroup life)": float(np.corrcoef(us, life)[0, 1]),
       "slow_chunks (> 1.06 x median)": {"share": float(slow.mean()), "clock_ghz": float(per_chunk[slow].mean()) if slow.any() else None, "time_us": float(us[slow].mean()) if slow.any() else None},
       "other_chunks": {"clock_ghz": float(per_chunk[~slow].mean()), "time_us": float(us[~slow].mean())},
       "series_us": [round(float(x), 2) for x in us], "series_ghz": [round(float(x), 3) for x in per_chunk],
       "note": "one small copy per 10 launches (the clock ring)"}
print(json.dumps(out))
