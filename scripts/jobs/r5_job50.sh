# single-env drop-ins through one pinned host image: the new test, the tests of the single-env surface, steps per second
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host_mirror or single_env or g1_simulator or g2_rhs or rk45_simulator" > gpurun_out/r5_single_env_tests.log 2>&1 || { tail -40 gpurun_out/r5_single_env_tests.log; exit 1; }
tail -3 gpurun_out/r5_single_env_tests.log
timeout -k 10 300 python -m pytest tests/test_sb3_contract.py -x -q -m gpu > gpurun_out/r5_single_env_sb3.log 2>&1 || { tail -40 gpurun_out/r5_single_env_sb3.log; exit 1; }
tail -2 gpurun_out/r5_single_env_sb3.log
timeout -k 10 300 python scripts/single_env_rate.py > gpurun_out/r5_single_env_rate.json 2> gpurun_out/r5_single_env_rate.err
cat gpurun_out/r5_single_env_rate.json
