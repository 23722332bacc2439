#!/bin/bash
# GPU box: rocprofv3 kernel-trace/stats of the rollout-side benchmarks -> gpurun_out/profile_policy/summary.json
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_policy
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/policy -- python3 $GRAFT_REPO_ROOT/scripts/policy_bench.py > $OUT/policy_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rollout -- python3 $GRAFT_REPO_ROOT/scripts/rollout_bench.py > $OUT/rollout_bench.log 2>&1
python3 - <<PY
import csv, glob, json
out = {}
for tag in ("policy", "rollout"):
    f = glob.glob("$OUT/%s/**/*kernel_stats.csv" % tag, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    out[tag + "_kernel_stats_top"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:8]]
    out[tag + "_bench_lines"] = [l.strip() for l in open("$OUT/%s_bench.log" % tag) if l.startswith("arch")]
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k in ("policy_bench_lines", "rollout_bench_lines"): print("\n".join(out[k]))
for r in out["policy_kernel_stats_top"][:4]: print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
