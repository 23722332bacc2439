set -e
L=cur=rl_rocket_6dof_amd/csrc/libr6dof.so
for N in 4194304 6291456 8388608 12582912; do
  timeout -k 10 400 python scripts/ab_swap.py --n $N --libs $L --shapes "tpw=2;tpw=3;tpw=4;tpw=5;tpw=8" --envs 2 --rounds 9 --steps 40 --burnin 600 --pool 8 > gpurun_out/r4_tpw2_n$N.log 2>&1
  echo N=$N; grep POOLED gpurun_out/r4_tpw2_n$N.log | cut -c1-120
done
