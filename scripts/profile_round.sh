#!/bin/bash
# GPU box: rocprofv3 kernel-trace/stats of the default bench command + PMC traffic passes -> gpurun_out/profile_<tag>/
set -e
TAG=${1:-r2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_under_trace.log 2>&1   # the default bench command
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --no-cpu-baseline --no-extras > $OUT/pmc_$c.log 2>&1
done
python3 - <<PY
import csv, glob, json
out = {}
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out["kernel_stats_top"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:6]]
tr = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
allk = list(csv.DictReader(open(tr)))
# the headline instance: RK4, acceleration shaping, 14-wide obs, no NT stores, single sub-step; the headline's timed
# region = its last 2000 launches before the first launch of the out_of_cache side measurement (NT instance)
name = lambda r: r["Kernel_Name"]
head = [i for i, r in enumerate(allk) if "step_kernel<1, 0, 14, false, true>" in name(r) or "step_kernelILi1ELi0ELi14ELb0ELb1E" in name(r)]
nt_first = next((i for i, r in enumerate(allk) if "step_kernel<1, 0, 14, true" in name(r) or "step_kernelILi1ELi0ELi14ELb1E" in name(r)), len(allk))
st = [allk[i] for i in head if i < nt_first]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in st][-2000:]   # = the timed region (bench.py default --steps)
out["step_kernel_timed_region"] = {"launches": len(d), "avg_us": sum(d) / len(d), "min_us": min(d), "max_us": max(d),
                                   "vgpr": st[-1]["VGPR_Count"], "sgpr": st[-1]["SGPR_Count"], "lds": st[-1]["LDS_Block_Size"],
                                   "grid": st[-1]["Grid_Size_X"], "workgroup": st[-1]["Workgroup_Size_X"]}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c][-50:]   # --no-extras: only the headline instance runs
    out[c + "_KB_per_launch_raw"] = sum(v) / len(v)
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B read requests at 64 B -> double it; WRITE_SIZE exact
fetch = out["FETCH_SIZE_KB_per_launch_raw"] * 1024 * 2
write = out["WRITE_SIZE_KB_per_launch_raw"] * 1024
out["hbm_bytes_per_launch"] = fetch + write
out["hbm_read_bytes_per_launch_corrected"] = fetch
out["hbm_write_bytes_per_launch"] = write
out["algorithmic_bytes_per_launch"] = 189 * (1 << 20)
out["commit"] = (open("$GRAFT_REPO_ROOT/rl_rocket_6dof_amd/csrc/BUILD_COMMIT").read().strip() if __import__("os").path.exists("$GRAFT_REPO_ROOT/rl_rocket_6dof_amd/csrc/BUILD_COMMIT") else "unknown")
by = {}
for r in allk:
    k = name(r)[:70]
    by.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out["kernels_in_the_run"] = {k: {"launches": len(v), "avg_us": sum(v) / len(v)} for k, v in by.items() if "r6" in k}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -1 $OUT/bench_under_trace.log
