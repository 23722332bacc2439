# PPO minibatch gradient with the SLP vectoriser on (v_pk_fma_f32 / v_pk_mul_f32 in a one-wave-per-SIMD kernel, where a packed instruction issues in 2.7 ns against 2 x 2.35)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 300 python scripts/ppo_grad_ab.py base=rl_rocket_6dof_amd/csrc/libr6dof.so slp=$V/libr6dof_slp.so > gpurun_out/r5_ppo_slp.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py base=rl_rocket_6dof_amd/csrc/libr6dof.so slp=$V/libr6dof_slp.so --sep >> gpurun_out/r5_ppo_slp.log 2>&1
cat gpurun_out/r5_ppo_slp.log
