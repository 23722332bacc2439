# round 5, job 2: are the slow 20-launch chunks data or chip?  + packed-fp32 issue rates
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 scripts/burst_probe.py > gpurun_out/r5_burst_probe.json 2> gpurun_out/r5_burst_probe.err
grep -v "^/opt" gpurun_out/r5_burst_probe.err | tail
timeout -k 10 120 scripts/pk_rate_bench > gpurun_out/r5_pk_rate_bench.log 2>&1
cat gpurun_out/r5_pk_rate_bench.log
