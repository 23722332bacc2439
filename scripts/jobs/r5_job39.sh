# timing-only build: dW2's f16 products accumulate straight into the launch's sums (no per-tile block, no join; the scale is wrong)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
L="base=rl_rocket_6dof_amd/csrc/libr6dof.so direct=$V/libr6dof_direct.so"
timeout -k 10 300 python scripts/ppo_grad_ab.py $L > gpurun_out/r5_ppo_direct.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --n=1048576 >> gpurun_out/r5_ppo_direct.log 2>&1
grep -v amdgpu.ids gpurun_out/r5_ppo_direct.log
