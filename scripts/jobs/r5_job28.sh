set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/timed_region_probe.py > gpurun_out/r5_timed_region_probe.json 2> gpurun_out/r5_timed_region_probe.err || { tail -5 gpurun_out/r5_timed_region_probe.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r5_timed_region_probe.json'))
for k,v in d.items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if a!='values'})"
