set -e
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 500 python scripts/ab_swap.py --libs soa=$V/libr6dof_untiled.so,tiled=$V/libr6dof_tiled.so --reset-on-swap 600 --envs 3 --rounds 9 --burnin 10 > gpurun_out/r4_swap_tiled.log 2>&1; grep POOLED gpurun_out/r4_swap_tiled.log | cut -c1-120
timeout -k 10 500 python scripts/ab_swap.py --n 8388608 --libs soa=$V/libr6dof_untiled.so,tiled=$V/libr6dof_tiled.so --reset-on-swap 400 --envs 1 --rounds 5 --steps 40 --burnin 10 --pool 8 > gpurun_out/r4_swap_tiled_8m.log 2>&1; grep POOLED gpurun_out/r4_swap_tiled_8m.log | cut -c1-120
timeout -k 10 500 python scripts/ab_swap.py --n 2097152 --libs soa=$V/libr6dof_untiled.so,tiled=$V/libr6dof_tiled.so --reset-on-swap 400 --envs 1 --rounds 5 --steps 60 --burnin 10 --pool 4 > gpurun_out/r4_swap_tiled_2m.log 2>&1; grep POOLED gpurun_out/r4_swap_tiled_2m.log | cut -c1-120
