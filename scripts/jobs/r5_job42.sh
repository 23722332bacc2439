# is MODE.FP16_OVFL what the direct form's missing gain went to?  (noovfl: the same kernel without the s_setreg)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
L="direct=rl_rocket_6dof_amd/csrc/libr6dof.so noovfl=$V/libr6dof_noovfl.so nodirect=$V/libr6dof_nodirect.so"
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --n=1048576 > gpurun_out/r5_ppo_ovfl.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L >> gpurun_out/r5_ppo_ovfl.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --sep >> gpurun_out/r5_ppo_ovfl.log 2>&1
grep -v amdgpu.ids gpurun_out/r5_ppo_ovfl.log
