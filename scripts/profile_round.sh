#!/bin/bash
# GPU box: rocprofv3 kernel-trace/stats of the default bench command + PMC traffic passes -> gpurun_out/profile_<tag>/
set -e
TAG=${1:-r2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_under_trace.log 2>&1   # the default bench command
# ... and the DRIVER's command shape (round-4 review, Next 1): 5 warm-up + 20 timed launches behind the 1 s pre-roll
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_driver -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_shape_under_trace.log 2>&1
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
allk = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
# the headline instance: RK4, acceleration shaping, 14-wide obs, no NT stores, single sub-step; the headline's timed
# region = its last 2000 launches before the first launch of the out_of_cache side measurement (NT instance)
name = lambda r: r["Kernel_Name"]
head = [i for i, r in enumerate(allk) if "step_kernel<1, 0, 14, false, true>" in name(r) or "step_kernelILi1ELi0ELi14ELb0ELb1E" in name(r)]
nt_first = next((i for i, r in enumerate(allk) if "step_kernel<1, 0, 14, true" in name(r) or "step_kernelILi1ELi0ELi14ELb1E" in name(r)), len(allk))
st = [allk[i] for i in head if i < nt_first]
dur = lambda rows: [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
def stats(d):
    s = sorted(d)
    return {"launches": len(d), "avg_us": sum(d) / len(d), "median_us": s[len(s) // 2], "p10_us": s[len(s) // 10], "p90_us": s[9 * len(s) // 10], "min_us": s[0], "max_us": s[-1]}
def chunk_means(d, c=20):
    m = sorted(sum(d[i:i + c]) / c for i in range(0, len(d) - c + 1, c))
    return {"chunks": len(m), "median_us": m[len(m) // 2], "p10_us": m[len(m) // 10], "p90_us": m[9 * len(m) // 10], "min_us": m[0], "max_us": m[-1],
            "share_above_1.06_x_median": sum(1 for x in m if x > 1.06 * m[len(m) // 2]) / len(m)}
# bench.py (default): burn-in 2000, 1 s pre-roll, 200 warm-up, 2000 TIMED, 2000 steady launches -- all of the headline instance, all before the first
# launch of the out_of_cache side measurement (NT instance)
d_all = dur(st)
d = d_all[-4000:-2000]
d_steady = d_all[-2000:]
out["step_kernel_timed_region"] = dict(stats(d), vgpr=st[-1]["VGPR_Count"], sgpr=st[-1]["SGPR_Count"], lds=st[-1]["LDS_Block_Size"],
                                       grid=st[-1]["Grid_Size_X"], workgroup=st[-1]["Workgroup_Size_X"],
                                       note="the 2000 launches of bench.py's timed region (default command), kernel durations from the trace")
out["step_kernel_steady_record"] = dict(stats(d_steady), chunk_means_of_20=chunk_means(d_steady), note="the 2000 launches of bench.py's `steady` record right behind the timed region")
out["step_kernel_preroll"] = {"launches": len(d_all) - 4200, "first_200_avg_us": sum(d_all[:200]) / 200, "avg_us_by_tenth": [sum(d_all[i:i + (len(d_all) - 4200) // 10]) / ((len(d_all) - 4200) // 10)
                                                                                                                        for i in range(0, (len(d_all) - 4200) // 10 * 10, (len(d_all) - 4200) // 10)]}
# the driver's command shape: burn-in 2000, pre-roll, 5 warm-up, 20 TIMED, 2000 steady
trd = glob.glob("$OUT/trace_driver/**/*kernel_trace.csv", recursive=True)
if trd:
    rows = list(csv.DictReader(open(trd[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    hd = [i for i, r in enumerate(rows) if "step_kernel<1, 0, 14, false, true>" in name(r) or "step_kernelILi1ELi0ELi14ELb0ELb1E" in name(r)]
    ntf = next((i for i, r in enumerate(rows) if "step_kernel<1, 0, 14, true" in name(r) or "step_kernelILi1ELi0ELi14ELb1E" in name(r)), len(rows))
    sd = [rows[i] for i in hd if i < ntf]
    dd = dur(sd)
    t20, s2000 = dd[-2020:-2000], dd[-2000:]
    span = (int(sd[-2001]["End_Timestamp"]) - int(sd[-2020]["Start_Timestamp"])) / 1e3 / 20
    line = [l for l in open("$OUT/bench_driver_shape_under_trace.log") if l.startswith("{")]
    bl = json.loads(line[-1]) if line else {}
    out["driver_shape_steps20_warmup5"] = {"timed_20_kernel_durations_us": [round(x, 2) for x in t20], "timed_20_mean_us": sum(t20) / 20,
                                           "timed_20_first_start_to_last_end_per_step_us": span,
                                           "steady_2000": dict(stats(s2000), chunk_means_of_20=chunk_means(s2000)),
                                           "bench_line": {"ms_per_step_us": bl.get("ms_per_step", 0) * 1e3, "kernel_us_hip_events": bl.get("roofline", {}).get("kernel_us"),
                                                          "frac": bl.get("roofline", {}).get("frac"), "steady": bl.get("steady")},
                                           "command": "rocprofv3 --kernel-trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    per = {}     # one value per dispatch (a counter may come as one row per XCD: summed)
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
            per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    v = [per[k] for k in sorted(per)][-50:]   # --no-extras: only the headline instance runs (the last launches are the `steady` record's)
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
# the raw per-launch tables (tens of MB with the 1 s pre-roll in the trace) stay on the box: summary.json and kernel_stats.csv are what is judged
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
