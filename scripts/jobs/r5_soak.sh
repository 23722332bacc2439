# round 5: soaks of the kernels that changed -- the adaptive step with guests (2^20 envs and 2^22 envs: list recycling on the balanced grid), r6_step_k's pair kernel, the RK4 / Euler step
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python scripts/soak.py 20000 rk45 > gpurun_out/r5_soak_rk45.log 2>&1; tail -2 gpurun_out/r5_soak_rk45.log
timeout -k 10 300 python scripts/stepk_soak.py > gpurun_out/r5_soak_stepk.log 2>&1; tail -2 gpurun_out/r5_soak_stepk.log
timeout -k 10 300 python scripts/soak.py 60000 rk4 > gpurun_out/r5_soak_rk4.log 2>&1; tail -2 gpurun_out/r5_soak_rk4.log
timeout -k 10 300 python scripts/soak.py 60000 euler > gpurun_out/r5_soak_euler.log 2>&1; tail -2 gpurun_out/r5_soak_euler.log
