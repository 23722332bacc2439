# PPO gradient launch (and the policy / rollout units) compiled with -mllvm -amdgpu-mfma-vgpr-form: MFMA results that the VALU consumes land in
# vector registers instead of being copied out of the accumulation file (v_accvgpr_read 505 -> 253 in the shared-trunk instance)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
L="base=rl_rocket_6dof_amd/csrc/libr6dof.so vform=$V/libr6dof_vform.so"
timeout -k 10 300 python scripts/ppo_grad_ab.py $L > gpurun_out/r5_ppo_vform.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --sep >> gpurun_out/r5_ppo_vform.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --n=1048576 >> gpurun_out/r5_ppo_vform.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --n=1048576 --sep >> gpurun_out/r5_ppo_vform.log 2>&1
grep -v amdgpu.ids gpurun_out/r5_ppo_vform.log
