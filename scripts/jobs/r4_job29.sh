set -e
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 400 python scripts/ab_swap.py --libs sp0=$V/libr6dof_sp0.so,sp1=$V/libr6dof_sp1.so --envs 2 --rounds 11 > gpurun_out/r4_swap_sp.log 2>&1; grep POOLED gpurun_out/r4_swap_sp.log | cut -c1-120
timeout -k 10 400 python scripts/ab_swap.py --libs sp0=$V/libr6dof_sp0.so,sp1=$V/libr6dof_sp1.so --envs 1 --rounds 7 --integrator euler > gpurun_out/r4_swap_sp_e.log 2>&1; grep POOLED gpurun_out/r4_swap_sp_e.log | cut -c1-120
