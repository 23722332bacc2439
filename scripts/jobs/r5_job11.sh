set -e
cd $GRAFT_REPO_ROOT
for N in 4194304 8388608; do
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --n $N --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so --shapes 'defer=0;defer=6;grid=1024,defer=0;grid=1024,defer=6' --envs 1 --rounds 7 --steps 30 --burnin 300 > gpurun_out/r5_swap_rk45_grid_$N.log 2>&1
echo N=$N; grep POOLED gpurun_out/r5_swap_rk45_grid_$N.log | cut -c1-160
done
