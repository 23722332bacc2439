set -e
V=rl_rocket_6dof_amd/csrc/variants
L="hot0=$V/libr6dof_hot0.so,old=$V/libr6dof_drain0.so,pre=$V/libr6dof_hot1.so,nopre=$V/libr6dof_hot1nopre.so,rot=$V/libr6dof_rot.so"
python scripts/ab_swap.py --libs $L --envs 3 > gpurun_out/r4_swap_rk4.log 2>&1
grep POOLED gpurun_out/r4_swap_rk4.log
python scripts/ab_swap.py --libs $L --envs 2 --integrator euler --rounds 11 > gpurun_out/r4_swap_euler.log 2>&1
grep POOLED gpurun_out/r4_swap_euler.log
python scripts/ab_swap.py --libs $L --envs 2 --wrapped --rounds 11 > gpurun_out/r4_swap_wrapped.log 2>&1
grep POOLED gpurun_out/r4_swap_wrapped.log
R="w1=$V/libr6dof_rk45w1.so,w3=$V/libr6dof_rk45w3.so,w4=$V/libr6dof_rk45w4.so,old=$V/libr6dof_drain0.so"
python scripts/ab_swap.py --libs $R --envs 1 --integrator rk45 --rounds 7 --steps 50 --burnin 600 > gpurun_out/r4_swap_rk45.log 2>&1
grep POOLED gpurun_out/r4_swap_rk45.log
