# round 5, job 7: RK45 after the register diet (one bookkeeping word, parked post-phase values, scalar min_step): parity + timing
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rk45 or g10 or g11" > gpurun_out/r5_rk45_tests.log 2>&1 || { tail -40 gpurun_out/r5_rk45_tests.log; exit 1; }
tail -1 gpurun_out/r5_rk45_tests.log
grep "nfev identical" gpurun_out/r5_rk45_tests.log || true
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so --shapes 'defer=0;defer=6' --envs 2 --rounds 9 --steps 60 --burnin 600 > gpurun_out/r5_swap_rk45_diet.log 2>&1
grep POOLED gpurun_out/r5_swap_rk45_diet.log | cut -c1-160
