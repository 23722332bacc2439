# the PPO gradient launch under -mllvm -amdgpu-sched-strategy=max-ilp and -amdgpu-use-amdgpu-trackers (every unit recompiled)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
L="base=rl_rocket_6dof_amd/csrc/libr6dof.so maxilp=$V/libr6dof_pmaxilp.so trackers=$V/libr6dof_ptrk.so"
timeout -k 10 300 python scripts/ppo_grad_ab.py $L > gpurun_out/r5_ppo_sched.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --sep >> gpurun_out/r5_ppo_sched.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --n=1048576 --sep >> gpurun_out/r5_ppo_sched.log 2>&1
grep -v amdgpu.ids gpurun_out/r5_ppo_sched.log
