# round 5, job 3: slow chunks -- replay with the data or not; sensors meanwhile
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 scripts/burst_probe2.py > gpurun_out/r5_burst_probe2.json 2> gpurun_out/r5_burst_probe2.err
grep -v "^/opt" gpurun_out/r5_burst_probe2.err | tail -20
