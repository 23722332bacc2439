# the step kernels under other instruction-scheduling strategies of the compiler (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause, -amdgpu-use-amdgpu-trackers)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 600 python scripts/ab_swap.py --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so,maxilp=$V/libr6dof_maxilp.so,trackers=$V/libr6dof_trk.so,memclause=$V/libr6dof_memcl.so --envs 2 --rounds 15 --steps 200 --reset-on-swap 300 > gpurun_out/r5_swap_sched.log 2>&1
grep POOLED gpurun_out/r5_swap_sched.log | cut -c1-160
