set -e
L=cur=rl_rocket_6dof_amd/csrc/libr6dof.so
timeout -k 10 500 python scripts/ab_swap.py --libs $L --integrator rk45 --shapes "base;even=0.5;grid=4096;grid=2048;grid=8192;even=0.5,flags=4;grid=4096,flags=4" --envs 1 --rounds 7 --steps 50 --burnin 600 > gpurun_out/r4_swap_rk45c.log 2>&1
grep POOLED gpurun_out/r4_swap_rk45c.log | cut -c1-120
