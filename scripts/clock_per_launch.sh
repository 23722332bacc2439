#!/bin/bash
# GPU box: effective shader clock of every step launch -- GRBM_GUI_ACTIVE (gfx-clock cycles the GPU was active) over the launch's
# duration -- to tell a slow launch that took more CYCLES from one that ran at a lower CLOCK.  usage: scripts/clock_per_launch.sh <tag>
set -e
TAG=${1:-r5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/clock_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3000 --warmup 200 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, json
import numpy as np
cc = glob.glob("$OUT/run/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(cc)) if "step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
print("columns:", list(rows[0].keys()))
tr = glob.glob("$OUT/run/**/*kernel_trace.csv", recursive=True)
dur = {}
if tr:
    for r in csv.DictReader(open(tr[0])):
        dur[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
recs = []
for r in rows:
    st, en = (int(r["Start_Timestamp"]), int(r["End_Timestamp"])) if "Start_Timestamp" in r and r["Start_Timestamp"] else dur.get(r["Dispatch_Id"], (0, 0))
    if en > st:
        recs.append((st, (en - st) / 1e3, float(r["Counter_Value"])))
recs.sort()
recs = recs[2200:]      # after the burn-in
d = np.array([x[1] for x in recs]); c = np.array([x[2] for x in recs])
# GRBM_GUI_ACTIVE is summed over the 8 XCDs' GRBMs (or not): report cycles / duration as is, normalised by its median
clk = c / d   # cycles per us
out = {"launches": len(recs), "duration_us": {"median": float(np.median(d)), "p10": float(np.percentile(d, 10)), "p90": float(np.percentile(d, 90))},
       "cycles_per_us_raw": {"median": float(np.median(clk)), "p10": float(np.percentile(clk, 10)), "p90": float(np.percentile(clk, 90))},
       "cycles": {"median": float(np.median(c)), "p10": float(np.percentile(c, 10)), "p90": float(np.percentile(c, 90))},
       "corr_duration_cycles": float(np.corrcoef(d, c)[0, 1]), "corr_duration_clock": float(np.corrcoef(d, clk)[0, 1])}
slow = d > np.median(d) * 1.06
out["slow_launches"] = {"share": float(slow.mean()), "cycles_rel_to_median": float(c[slow].mean() / np.median(c)) if slow.any() else None,
                        "clock_rel_to_median": float(clk[slow].mean() / np.median(clk)) if slow.any() else None,
                        "duration_rel_to_median": float(d[slow].mean() / np.median(d)) if slow.any() else None}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
