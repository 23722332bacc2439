set -e
V=rl_rocket_6dof_amd/csrc/variants
# (1) 2^20: tiles per wave made even (16384 tiles = 4096 waves x 4): grid of 1024 workgroups, with / without the XCD split; prefetching 4-wave build
python scripts/ab_swap.py --libs hot0=$V/libr6dof_hot0.so,pf4=$V/libr6dof_pf4.so,pf5=$V/libr6dof_pf5.so --shapes "base;grid=1024;grid=1024,even=0.5;grid=1024,even=0.52;even=0.5" --envs 2 --rounds 15 > gpurun_out/r4_swap_grid.log 2>&1
grep POOLED gpurun_out/r4_swap_grid.log
# (2) large N
for N in 2097152 4194304 8388608; do
  python scripts/ab_swap.py --n $N --libs hot0=$V/libr6dof_hot0.so,nt2m=$V/libr6dof_nt2m.so,pf5=$V/libr6dof_pf5.so,pf4=$V/libr6dof_pf4.so,pf4nt=$V/libr6dof_pf4nt.so --envs 1 --rounds 9 --steps 60 --burnin 600 --pool 4 > gpurun_out/r4_swap_n$N.log 2>&1
  echo N=$N; grep POOLED gpurun_out/r4_swap_n$N.log
done
# (3) RK45: fast step-control arithmetic at 3 / 4 waves, against round 3's kernel (old)
python scripts/ab_swap.py --libs old=$V/libr6dof_drain0.so,w3=$V/libr6dof_rk45w3.so,fw3=$V/libr6dof_rk45fw3.so,fw4=$V/libr6dof_rk45fw4.so --envs 1 --integrator rk45 --rounds 7 --steps 50 --burnin 600 > gpurun_out/r4_swap_rk45b.log 2>&1
grep POOLED gpurun_out/r4_swap_rk45b.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "rk45" 2>&1 | grep -a "^\[\|passed\|failed" | cut -c1-250
