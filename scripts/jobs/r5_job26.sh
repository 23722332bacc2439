set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputest_3.log 2>&1 || { tail -60 gpurun_out/r5_gputest_3.log; exit 1; }
tail -1 gpurun_out/r5_gputest_3.log
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 600 python scripts/ab_swap.py --libs r5=rl_rocket_6dof_amd/csrc/libr6dof.so,r4=$V/libr6dof_r4.so --envs 3 --rounds 21 --steps 200 > gpurun_out/r5_swap_r4_vs_r5_rk4_final.log 2>&1
grep POOLED gpurun_out/r5_swap_r4_vs_r5_rk4_final.log | cut -c1-160
