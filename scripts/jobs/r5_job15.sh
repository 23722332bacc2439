set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_step_k.py -x -q > gpurun_out/r5_stepk_tests.log 2>&1 || { tail -40 gpurun_out/r5_stepk_tests.log; exit 1; }
tail -1 gpurun_out/r5_stepk_tests.log
timeout -k 10 300 python scripts/kstep_bench.py --n 65536 --k 4 16 64 > gpurun_out/r5_kstep_bench.log 2>&1
cat gpurun_out/r5_kstep_bench.log
