set -e
L=cur=rl_rocket_6dof_amd/csrc/libr6dof.so
timeout -k 10 600 python scripts/ab_swap.py --libs $L --shapes "base;even=0.5;even=0.52;even=0.53;even=0.55;even=0.56;even=0.58;even=0.6;grid=1216;grid=1152;sweep=1;sweep=2;flags=4;nt=1" --envs 3 --rounds 11 > gpurun_out/r4_swap_tune.log 2>&1
grep POOLED gpurun_out/r4_swap_tune.log | cut -c1-125
timeout -k 10 400 python scripts/ab_swap.py --libs $L --wrapped --shapes "base;even=0.5;even=0.52;even=0.56;even=0.58;flags=4;nt=1" --envs 2 --rounds 9 > gpurun_out/r4_swap_tune_w.log 2>&1
grep POOLED gpurun_out/r4_swap_tune_w.log | cut -c1-125
