# round 5, job 6: RK45 -- PMC counts with / without the straggler hand-over; workgroup-dynamic tile hand-out A/B
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so,wgdyn=$V/libr6dof_wgdyn.so --shapes 'defer=0;defer=6' --envs 2 --rounds 9 --steps 60 --burnin 600 > gpurun_out/r5_swap_rk45_wgdyn.log 2>&1
grep POOLED gpurun_out/r5_swap_rk45_wgdyn.log | cut -c1-160
bash scripts/pmc_rk45.sh r5 > gpurun_out/r5_pmc_rk45.log 2>&1 || { tail -20 gpurun_out/r5_pmc_rk45.log; exit 1; }
tail -3 gpurun_out/r5_pmc_rk45.log
