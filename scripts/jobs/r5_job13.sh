set -e
cd $GRAFT_REPO_ROOT
bash scripts/pmc_rk45.sh r5b > gpurun_out/r5_pmc_rk45b.log 2>&1 || { tail -20 gpurun_out/r5_pmc_rk45b.log; exit 1; }
tail -3 gpurun_out/r5_pmc_rk45b.log
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --n 4194304 --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so --shapes 'defer=6;tpw=4,defer=6;tpw=8,defer=6' --envs 1 --rounds 7 --steps 30 --burnin 300 > gpurun_out/r5_swap_rk45_tpw.log 2>&1
grep POOLED gpurun_out/r5_swap_rk45_tpw.log | cut -c1-160
