set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/clock_vs_duration.py > gpurun_out/r5_clock_vs_duration.json 2> gpurun_out/r5_clock_vs_duration.err || { tail -5 gpurun_out/r5_clock_vs_duration.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r5_clock_vs_duration.json')); print({k:v for k,v in d.items() if not k.startswith('series')})"
