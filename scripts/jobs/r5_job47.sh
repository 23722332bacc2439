# -mllvm -amdgpu-use-amdgpu-trackers on every instance: RK4 raw (3 envs x 21 rounds), Euler, fused stack, RK45, and r6_step_k (K = 4, 16, 64)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
L="--libs base=rl_rocket_6dof_amd/csrc/libr6dof.so,trackers=$V/libr6dof_trk.so"
timeout -k 10 600 python scripts/ab_swap.py $L --envs 3 --rounds 21 --steps 200 > gpurun_out/r5_swap_trk_rk4.log 2>&1
grep POOLED gpurun_out/r5_swap_trk_rk4.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --integrator euler $L --envs 2 --rounds 15 --steps 200 > gpurun_out/r5_swap_trk_euler.log 2>&1
grep POOLED gpurun_out/r5_swap_trk_euler.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --wrapped $L --envs 2 --rounds 15 --steps 200 > gpurun_out/r5_swap_trk_wrapped.log 2>&1
grep POOLED gpurun_out/r5_swap_trk_wrapped.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 $L --envs 2 --rounds 9 --steps 60 --burnin 600 > gpurun_out/r5_swap_trk_rk45.log 2>&1
grep POOLED gpurun_out/r5_swap_trk_rk45.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --n 8388608 $L --envs 1 --rounds 9 --steps 40 --burnin 300 > gpurun_out/r5_swap_trk_2p23.log 2>&1
grep POOLED gpurun_out/r5_swap_trk_2p23.log | cut -c1-160
for lib in rl_rocket_6dof_amd/csrc/libr6dof.so $V/libr6dof_trk.so rl_rocket_6dof_amd/csrc/libr6dof.so $V/libr6dof_trk.so; do
  echo $lib >> gpurun_out/r5_kstep_trk.log
  timeout -k 10 300 python scripts/kstep_bench.py --n 65536 --k 4 16 64 --lib $GRAFT_REPO_ROOT/$lib 2>&1 | grep "K=" >> gpurun_out/r5_kstep_trk.log
done
cat gpurun_out/r5_kstep_trk.log
