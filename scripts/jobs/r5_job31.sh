set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "groups" > gpurun_out/r5_groups_test.log 2>&1 || { tail -30 gpurun_out/r5_groups_test.log; exit 1; }
tail -1 gpurun_out/r5_groups_test.log
