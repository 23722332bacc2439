set -e
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 500 python scripts/ab_swap.py --libs cur=rl_rocket_6dof_amd/csrc/libr6dof.so,dyn=$V/libr6dof_rk45dyn.so --integrator rk45 --shapes "base;even=0.51;even=0.52;even=0.48" --envs 1 --rounds 7 --steps 50 --burnin 600 > gpurun_out/r4_swap_rk45d.log 2>&1
grep POOLED gpurun_out/r4_swap_rk45d.log | cut -c1-120
timeout -k 10 300 python scripts/launch_shape_ab.py --libs cur=rl_rocket_6dof_amd/csrc/libr6dof.so,dyn=$V/libr6dof_rk45dyn.so --integrator rk45 --n 262144 --shapes base --rounds 1 --steps 10 --burnin 10 > gpurun_out/r4_rk45dyn_check.log 2>&1
grep bit-exact gpurun_out/r4_rk45dyn_check.log
