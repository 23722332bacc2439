set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 600 python scripts/ab_swap.py --libs r5=rl_rocket_6dof_amd/csrc/libr6dof.so,r4=$V/libr6dof_r4.so --envs 3 --rounds 21 --steps 200 > gpurun_out/r5_swap_r4_vs_r5_rk4.log 2>&1
grep POOLED gpurun_out/r5_swap_r4_vs_r5_rk4.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --integrator euler --libs r5=rl_rocket_6dof_amd/csrc/libr6dof.so,r4=$V/libr6dof_r4.so --envs 2 --rounds 15 --steps 200 > gpurun_out/r5_swap_r4_vs_r5_euler.log 2>&1
grep POOLED gpurun_out/r5_swap_r4_vs_r5_euler.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --wrapped --libs r5=rl_rocket_6dof_amd/csrc/libr6dof.so,r4=$V/libr6dof_r4.so --envs 2 --rounds 15 --steps 200 > gpurun_out/r5_swap_r4_vs_r5_wrapped.log 2>&1
grep POOLED gpurun_out/r5_swap_r4_vs_r5_wrapped.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --libs r5=rl_rocket_6dof_amd/csrc/libr6dof.so,r4=$V/libr6dof_r4.so --envs 2 --rounds 9 --steps 60 --burnin 600 > gpurun_out/r5_swap_r4_vs_r5_rk45.log 2>&1
grep POOLED gpurun_out/r5_swap_r4_vs_r5_rk45.log | cut -c1-160
