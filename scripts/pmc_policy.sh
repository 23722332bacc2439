#!/bin/bash
# GPU box: PMC passes for the fused policy kernel (policy_bench.py at N = 2^20), one counter group per run
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_policy
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/scripts/policy_bench.py > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections
res = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "policy_act_kernelILi64ELi64" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in by.items():
        v = v[-20:]
        res[k] = sum(v) / len(v)
for k, v in res.items(): print(f"{k:28s} {v:18.1f}")
PY
