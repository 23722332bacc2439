# the closed-loop rollout unit with -mllvm -amdgpu-mfma-vgpr-form (registers 248 -> 171 for (64,64), 304 -> 236 for [128,64]): two processes on one box
set -e
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rl_rocket_6dof_amd/csrc/variants
for rep in 1 2; do
for v in base vform; do
  if [ $v == base ]; then export R6_LIB=$GRAFT_REPO_ROOT/rl_rocket_6dof_amd/csrc/libr6dof.so; else export R6_LIB=$V/libr6dof_vform.so; fi
  echo "== $v (rep $rep)" >> gpurun_out/r5_rollout_vform.log
  timeout -k 10 300 python scripts/rollout_fused_bench.py --n 16384 65536 1048576 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5_rollout_vform.log
done; done
cat gpurun_out/r5_rollout_vform.log
