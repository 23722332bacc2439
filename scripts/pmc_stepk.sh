#!/bin/bash
# usage: scripts/pmc_stepk.sh <tag> [driver args...]   (GPU box) -- PMC passes for r6::step_k_kernel at BASELINE configs[1]
# (65 536 envs, Euler dt 0.04, K = 16), one counter group per run, --kernel-trace only.
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_stepk_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/scripts/stepk_driver.py "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $OUT/g$i.log)"
done
python3 - <<PY
import csv, glob, collections, json
res = collections.OrderedDict()
dur = []
for f in sorted(glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "step_k_" in r["Kernel_Name"]]
    by = collections.defaultdict(dict)     # counter -> dispatch -> value (rows of one dispatch are summed)
    for r in rows:
        d = by[r["Counter_Name"]]
        d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    for k, d in by.items():
        v = [d[i] for i in sorted(d)][-30:]
        res[k] = sum(v) / len(v)
for f in sorted(glob.glob("$OUT/g1/**/*kernel_trace.csv", recursive=True)):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "step_k_" in r["Kernel_Name"]][-30:]
    if d: res["launch_us_under_counters"] = sum(d) / len(d)
res["driver_line"] = open("$OUT/g1.log").read().strip().split("\n")[-1]
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, v in res.items(): print(f"{k:28s} {v}")
PY
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
