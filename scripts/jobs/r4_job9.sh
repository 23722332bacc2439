set -e
L=cur=rl_rocket_6dof_amd/csrc/libr6dof.so
# bit-exactness of the hand-out first (300 steps from reset, raw and wrapped), then timing on one set of buffers
timeout -k 10 300 python scripts/launch_shape_ab.py --libs $L --shapes "base;dyn=0.06;dyn=0.12;dyn=0.2" --rounds 3 --steps 50 > gpurun_out/r4_dyn_check.log 2>&1
grep "bit-exact" gpurun_out/r4_dyn_check.log
timeout -k 10 300 python scripts/launch_shape_ab.py --libs $L --shapes "base;dyn=0.12" --rounds 3 --steps 50 --wrapped > gpurun_out/r4_dyn_check_w.log 2>&1
grep "bit-exact" gpurun_out/r4_dyn_check_w.log
timeout -k 10 400 python scripts/ab_swap.py --libs $L --shapes "base;dyn=0.03;dyn=0.06;dyn=0.09;dyn=0.12;dyn=0.16;dyn=0.2;dyn=0.12,even=0.5;dyn=0.2,even=0.5;dyn=0.3,even=0.5" --envs 2 --rounds 15 > gpurun_out/r4_swap_dyn.log 2>&1
grep POOLED gpurun_out/r4_swap_dyn.log | cut -c1-120
timeout -k 10 300 python scripts/ab_swap.py --n 8388608 --libs $L --shapes "base;flags=4;tpw=4;tpw=4,flags=4" --envs 1 --rounds 7 --steps 40 --burnin 600 --pool 8 > gpurun_out/r4_swap_8m.log 2>&1
grep POOLED gpurun_out/r4_swap_8m.log | cut -c1-120
