# the step kernels under more code-generation switches of the compiler (only those whose code differs from the default build):
# -amdgpu-prealloc-sgpr-spill-vgprs, -enable-post-misched=0, -amdgpu-schedule-relaxed-occupancy, -greedy-regclass-priority-trumps-globalness,
# -amdgpu-set-wave-priority  (-amdgpu-max-memory-clause=32 and -amdgpu-schedule-metric-bias=100: byte-identical step kernels)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
L="base=rl_rocket_6dof_amd/csrc/libr6dof.so,prealloc=$V/libr6dof_prealloc.so,nopostra=$V/libr6dof_nopostra.so,relaxed=$V/libr6dof_relaxed.so,greedyrc=$V/libr6dof_greedyrc.so,setprio=$V/libr6dof_setprio.so"
timeout -k 10 600 python scripts/ab_swap.py --libs $L --envs 2 --rounds 15 --steps 200 > gpurun_out/r5_swap_cg_rk4.log 2>&1
grep POOLED gpurun_out/r5_swap_cg_rk4.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --libs $L --envs 2 --rounds 11 --steps 200 --integrator euler --dt 0.04 > gpurun_out/r5_swap_cg_euler.log 2>&1
grep POOLED gpurun_out/r5_swap_cg_euler.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --libs $L --envs 2 --rounds 9 --steps 100 --integrator rk45 > gpurun_out/r5_swap_cg_rk45.log 2>&1
grep POOLED gpurun_out/r5_swap_cg_rk45.log | cut -c1-160
