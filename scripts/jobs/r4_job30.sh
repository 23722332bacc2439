set -e
V=rl_rocket_6dof_amd/csrc/variants
for N in 8388608 16777216 2097152; do
timeout -k 10 400 python scripts/ab_swap.py --n $N --libs base=$V/libr6dof_ntl0.so,ntl=$V/libr6dof_ntl.so --envs 2 --rounds 7 --steps 40 --burnin 600 --pool 8 > gpurun_out/r4_swap_ntl_$N.log 2>&1; echo N=$N; grep POOLED gpurun_out/r4_swap_ntl_$N.log | cut -c1-120
done
