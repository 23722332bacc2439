set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so --shapes 'defer=0;defer=1;defer=2;defer=3;defer=4;defer=6;defer=8;defer=12' --envs 2 --rounds 9 --steps 60 --burnin 600 > gpurun_out/r5_swap_rk45_guests.log 2>&1
grep POOLED gpurun_out/r5_swap_rk45_guests.log | cut -c1-160
for N in 65536 262144 4194304; do
timeout -k 10 600 python scripts/ab_swap.py --integrator rk45 --n $N --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so --shapes 'defer=0;defer=6' --envs 1 --rounds 7 --steps 40 --burnin 300 > gpurun_out/r5_swap_rk45_guests_$N.log 2>&1
echo N=$N; grep POOLED gpurun_out/r5_swap_rk45_guests_$N.log | cut -c1-160
done
