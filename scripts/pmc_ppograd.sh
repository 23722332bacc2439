#!/bin/bash
# usage: scripts/pmc_ppograd.sh <tag>   (GPU box) -- PMC passes for r6u::ppo_grad_kernel (65 536-sample minibatches of the
# 16 384-env x 64-step training configuration, scripts/ppo_update_kernels.py), one counter group per run, --kernel-trace only.
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ppograd_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/scripts/ppo_update_kernels.py > $OUT/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $OUT/g$i.log)"
done
python3 - <<PY
import csv, glob, collections, json
res = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "ppo_grad_kernel" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in by.items():
        v = v[-30:]
        res[k] = sum(v) / len(v)
for f in sorted(glob.glob("$OUT/g1/**/*kernel_trace.csv", recursive=True)):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "ppo_grad_kernel" in r["Kernel_Name"]][-30:]
    if d: res["launch_us_under_counters"] = sum(d) / len(d)
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, v in res.items(): print(f"{k:28s} {v}")
PY
