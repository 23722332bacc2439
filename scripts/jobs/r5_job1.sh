# round 5, job 1: the driver's command shape on a fresh box -- bracket times, per-launch times, the kernel trace of the same command
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_driver_shape_1.json 2> gpurun_out/r5_driver_shape_1.err
python3 -c "import json; d=json.load(open('gpurun_out/r5_driver_shape_1.json')); print('driver shape 1:', d['ms_per_step']*1e3, d['roofline']['kernel_us'])"
for i in 2 3 4; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r5_driver_shape_$i.json 2> gpurun_out/r5_driver_shape_$i.err
python3 -c "import json; d=json.load(open('gpurun_out/r5_driver_shape_$i.json')); print('driver shape $i:', d['ms_per_step']*1e3, d['roofline']['kernel_us'])"
done
timeout -k 10 500 python3 scripts/driver_shape_probe.py > gpurun_out/r5_driver_shape_probe.json 2> gpurun_out/r5_driver_shape_probe.err
grep -v '^first' gpurun_out/r5_driver_shape_probe.err | tail -20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5_trace_driver_shape -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r5_trace_driver_shape.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json
tr = glob.glob("gpurun_out/r5_trace_driver_shape/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(tr)) if "step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-25:]
t0 = int(last[0]["Start_Timestamp"])
rec = [{"start_us": (int(r["Start_Timestamp"]) - t0) / 1e3, "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3} for r in last]
gaps = [rec[i + 1]["start_us"] - rec[i]["start_us"] - rec[i]["dur_us"] for i in range(24)]
burn = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[:-25]]
out = {"warmup5_and_timed20": rec, "gaps_us": gaps, "timed20_mean_dur_us": sum(r["dur_us"] for r in rec[5:]) / 20,
       "burnin_mean_us_by_200": [sum(burn[i:i + 200]) / len(burn[i:i + 200]) for i in range(0, len(burn), 200)]}
json.dump(out, open("gpurun_out/r5_trace_driver_shape_summary.json", "w"), indent=1)
print("trace: timed20 mean", out["timed20_mean_dur_us"], "durs", [round(r["dur_us"], 1) for r in rec], "gaps", [round(g, 1) for g in gaps])
print("burnin by 200:", [round(x, 1) for x in out["burnin_mean_us_by_200"]])
PY
tail -1 gpurun_out/r5_trace_driver_shape.log | cut -c1-300
