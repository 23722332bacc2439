set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/stepk_pair_trace.py --k 16 > gpurun_out/r5_stepk_pair_trace.log 2>&1
cat gpurun_out/r5_stepk_pair_trace.log | cut -c1-420
timeout -k 10 300 python scripts/kstep_bench.py --n 65536 --k 4 16 64 > gpurun_out/r5_kstep_bench.log 2>&1
cat gpurun_out/r5_kstep_bench.log
