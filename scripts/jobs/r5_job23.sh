set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "straggler" > gpurun_out/r5_rk45_guest_test.log 2>&1 || { tail -40 gpurun_out/r5_rk45_guest_test.log; exit 1; }
tail -2 gpurun_out/r5_rk45_guest_test.log
