set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_bench_launcher.py -x -q -m gpu > gpurun_out/r5_bench_launcher_tests.log 2>&1 || { tail -40 gpurun_out/r5_bench_launcher_tests.log; exit 1; }
tail -2 gpurun_out/r5_bench_launcher_tests.log
