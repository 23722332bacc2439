#!/bin/bash
# usage: scripts/pmc.sh <tag> [bench args...]   (GPU box) -- PMC passes for the step kernel, separate runs per counter group
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 300 --no-cpu-baseline "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections
res = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
    by = collections.defaultdict(dict)     # counter -> dispatch -> value (rows of one dispatch, e.g. one per XCD, are summed)
    for r in rows:
        d = by[r["Counter_Name"]]
        d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    for k, d in by.items():
        v = [d[i] for i in sorted(d)][-30:]
        res[k] = sum(v) / len(v)
for k, v in res.items(): print(f"{k:28s} {v:16.1f}")
import json
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
PY
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
