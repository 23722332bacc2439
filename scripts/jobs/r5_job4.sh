# round 5, job 4: RK45 straggler pass -- parity, bit-identity, A/B of the threshold; shader clock per launch of the headline kernel
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rk45 or g10 or g11" > gpurun_out/r5_rk45_tests.log 2>&1 || { tail -40 gpurun_out/r5_rk45_tests.log; exit 1; }
tail -3 gpurun_out/r5_rk45_tests.log
grep "nfev identical" gpurun_out/r5_rk45_tests.log || true
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so --shapes 'defer=0;defer=2;defer=4;defer=6;defer=10;defer=16' --envs 2 --rounds 9 --steps 60 --burnin 600 > gpurun_out/r5_swap_rk45_defer.log 2>&1
grep POOLED gpurun_out/r5_swap_rk45_defer.log | cut -c1-160
bash scripts/clock_per_launch.sh r5 > gpurun_out/r5_clock_per_launch.log 2>&1 || { tail -20 gpurun_out/r5_clock_per_launch.log; exit 1; }
tail -32 gpurun_out/r5_clock_per_launch.log
