# dW2's products straight into the launch's sums (R6U_DW2_DIRECT): accuracy against float64 autograd and time, against the form with a block per
# tile (nodirect = -DR6U_DW2_DIRECT=0) and with the first tile's largest delta2 entry placed at 2^3 instead of 2^-3 (sh10)
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
L="direct=rl_rocket_6dof_amd/csrc/libr6dof.so nodirect=$V/libr6dof_nodirect.so sh10=$V/libr6dof_sh10.so"
timeout -k 10 600 python scripts/ppo_grad_accuracy.py $L > gpurun_out/r5_ppo_direct_acc.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L > gpurun_out/r5_ppo_direct_ab.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --sep >> gpurun_out/r5_ppo_direct_ab.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --n=1048576 >> gpurun_out/r5_ppo_direct_ab.log 2>&1
timeout -k 10 300 python scripts/ppo_grad_ab.py $L --n=1048576 --sep >> gpurun_out/r5_ppo_direct_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r5_ppo_direct_acc.log | tail -40; grep -v amdgpu.ids gpurun_out/r5_ppo_direct_ab.log
timeout -k 10 600 python -m pytest tests/test_gpu_ppo.py -x -q -s > gpurun_out/r5_ppo_direct_test.log 2>&1 || { tail -30 gpurun_out/r5_ppo_direct_test.log; exit 1; }
grep "tiles 2^" gpurun_out/r5_ppo_direct_test.log; tail -2 gpurun_out/r5_ppo_direct_test.log
