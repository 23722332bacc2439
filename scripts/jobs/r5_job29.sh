set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --libs new=rl_rocket_6dof_amd/csrc/libr6dof.so,prev=$V/libr6dof_prev.so,r4=$V/libr6dof_r4.so --envs 2 --rounds 11 --steps 60 --burnin 600 > gpurun_out/r5_swap_rk45_zero.log 2>&1
grep POOLED gpurun_out/r5_swap_rk45_zero.log | cut -c1-160
