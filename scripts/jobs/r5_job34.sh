# PPO minibatch gradient: how much the dW1 products (exact fp32 16x16x4 MFMAs) and the VALU row sums cost (timing-only builds: wrong gradients)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 300 python scripts/ppo_grad_ab.py base=rl_rocket_6dof_amd/csrc/libr6dof.so no_dw1=$V/libr6dof_abl2.so no_dw1_dw3_db2=$V/libr6dof_abl7.so > gpurun_out/r5_ppo_ablate.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py base=rl_rocket_6dof_amd/csrc/libr6dof.so no_dw1=$V/libr6dof_abl2.so no_dw1_dw3_db2=$V/libr6dof_abl7.so --sep >> gpurun_out/r5_ppo_ablate.log 2>&1
cat gpurun_out/r5_ppo_ablate.log
