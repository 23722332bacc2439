#!/bin/bash
# GPU box: PMC passes of the adaptive step kernel with and without the straggler hand-over.  usage: scripts/pmc_rk45.sh <tag>
set -e
TAG=${1:-r5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_rk45_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for D in 0 6; do
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/d${D}_g$i -- python3 $GRAFT_REPO_ROOT/scripts/rk45_driver.py $D 30 > $OUT/d${D}_g$i.log 2>&1 || echo "defer $D group $i failed"
done
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for D in (0, 6):
    res = collections.OrderedDict()
    for f in sorted(glob.glob("$OUT/d%d_g*/**/*counter_collection.csv" % D, recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
        by = collections.defaultdict(lambda: collections.defaultdict(float))
        dur = {}
        for r in rows:
            by[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])      # summed over the XCDs' rows of a dispatch
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for k, v in by.items():
            ids = sorted(v, key=int)[-30:]
            res[k] = sum(v[i] for i in ids) / len(ids)
        ids = sorted(dur, key=int)[-30:]
        res["duration_us_under_pmc"] = sum(dur[i] for i in ids) / len(ids)
    tiles = (1 << 20) / 64
    res["derived"] = {"valu_per_tile": res.get("SQ_INSTS_VALU", 0) / tiles, "salu_per_tile": res.get("SQ_INSTS_SALU", 0) / tiles,
                      "lds_per_tile": res.get("SQ_INSTS_LDS", 0) / tiles,
                      "lanes_active_per_valu_instruction": res.get("SQ_THREAD_CYCLES_VALU", 0) / max(res.get("SQ_ACTIVE_INST_VALU", 1), 1) if "SQ_THREAD_CYCLES_VALU" in res else None}
    out["defer_%d" % D] = res
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for D, r in out.items():
    print(D, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
PY
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
