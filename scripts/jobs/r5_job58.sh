set -e
cd $GRAFT_REPO_ROOT
bash scripts/jobs/r5_job56.sh
unset R6_DIST_BACKEND R6_FORCE_DEVICE
timeout -k 10 1100 python -m pytest tests/test_gpu_bench_launcher.py -x -q -m gpu > gpurun_out/r5_bench_launcher_tests.log 2>&1 || { tail -40 gpurun_out/r5_bench_launcher_tests.log; exit 1; }
tail -1 gpurun_out/r5_bench_launcher_tests.log
