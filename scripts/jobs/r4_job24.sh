set -e
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 300 python scripts/launch_shape_ab.py --libs m0=$V/libr6dof_mid0.so,m1=$V/libr6dof_mid1.so,m2=$V/libr6dof_mid2.so --shapes base --rounds 1 --steps 20 --burnin 50 > gpurun_out/r4_mid_check.log 2>&1; grep bit-exact gpurun_out/r4_mid_check.log
timeout -k 10 400 python scripts/ab_swap.py --libs m0=$V/libr6dof_mid0.so,m1=$V/libr6dof_mid1.so,m2=$V/libr6dof_mid2.so --envs 2 --rounds 11 > gpurun_out/r4_swap_mid.log 2>&1; grep POOLED gpurun_out/r4_swap_mid.log | cut -c1-120
timeout -k 10 400 python scripts/ab_swap.py --n 8388608 --libs m0=$V/libr6dof_mid0.so,m1=$V/libr6dof_mid1.so,m2=$V/libr6dof_mid2.so --envs 1 --rounds 7 --steps 40 --burnin 600 --pool 8 > gpurun_out/r4_swap_mid_8m.log 2>&1; grep POOLED gpurun_out/r4_swap_mid_8m.log | cut -c1-120
