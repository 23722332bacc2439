set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/stepk_pair_trace.py --k 16 64 > gpurun_out/r5_stepk_pair_trace2.log 2>&1
cat gpurun_out/r5_stepk_pair_trace2.log | cut -c1-460
