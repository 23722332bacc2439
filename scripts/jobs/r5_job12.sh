set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rk45 or g10 or g11" > gpurun_out/r5_rk45_tests.log 2>&1 || { tail -40 gpurun_out/r5_rk45_tests.log; exit 1; }
tail -1 gpurun_out/r5_rk45_tests.log
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so --shapes 'defer=0;defer=6' --envs 2 --rounds 9 --steps 60 --burnin 600 > gpurun_out/r5_swap_rk45_recycle.log 2>&1
grep POOLED gpurun_out/r5_swap_rk45_recycle.log | cut -c1-160
for N in 2097152 4194304 8388608; do
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --n $N --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so --shapes 'defer=0;defer=6;grid=1024,defer=6' --envs 1 --rounds 7 --steps 30 --burnin 300 > gpurun_out/r5_swap_rk45_grid_$N.log 2>&1
echo N=$N; grep POOLED gpurun_out/r5_swap_rk45_grid_$N.log | cut -c1-160
done
