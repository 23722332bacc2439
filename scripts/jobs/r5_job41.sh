# does the time of the direct form depend on where the first tile's largest delta2 entry is placed (subnormal f16 halves in the matrix pipe / conversions)?
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
L="sh16=rl_rocket_6dof_amd/csrc/libr6dof.so sh10=$V/libr6dof_sh10.so sh4=$V/libr6dof_sh4.so sh0=$V/libr6dof_sh0.so nodirect=$V/libr6dof_nodirect.so"
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --n=1048576 > gpurun_out/r5_ppo_shift.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L >> gpurun_out/r5_ppo_shift.log 2>&1
grep -v amdgpu.ids gpurun_out/r5_ppo_shift.log
