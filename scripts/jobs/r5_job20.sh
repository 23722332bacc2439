set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_step_k.py tests/test_gpu_launch_shapes.py -x -q > gpurun_out/r5_quad_tests.log 2>&1 || { tail -40 gpurun_out/r5_quad_tests.log; exit 1; }
tail -1 gpurun_out/r5_quad_tests.log
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 600 python scripts/ab_swap.py --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so,noquad=$V/libr6dof_noquad.so --envs 2 --rounds 15 --steps 200 > gpurun_out/r5_swap_quad_rk4.log 2>&1
grep POOLED gpurun_out/r5_swap_quad_rk4.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --integrator euler --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so,noquad=$V/libr6dof_noquad.so --envs 2 --rounds 15 --steps 200 > gpurun_out/r5_swap_quad_euler.log 2>&1
grep POOLED gpurun_out/r5_swap_quad_euler.log | cut -c1-160
timeout -k 10 600 python scripts/ab_swap.py --wrapped --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so,noquad=$V/libr6dof_noquad.so --envs 2 --rounds 15 --steps 200 > gpurun_out/r5_swap_quad_wrapped.log 2>&1
grep POOLED gpurun_out/r5_swap_quad_wrapped.log | cut -c1-160
