set -e
L=cur=rl_rocket_6dof_amd/csrc/libr6dof.so
for N in 1310720 1572864 1835008 2097152 2621440 3145728 3407872 4194304 5242880 6291456; do
  python scripts/ab_swap.py --n $N --libs $L --shapes "nt=0;nt=1" --envs 1 --rounds 7 --steps 60 --burnin 600 --pool 4 > gpurun_out/r4_nt_n$N.log 2>&1
  echo N=$N; grep POOLED gpurun_out/r4_nt_n$N.log | cut -c1-110
done
for N in 4194304 8388608 16777216; do
  python scripts/ab_swap.py --n $N --libs $L --shapes "tpw=2;tpw=1;tpw=3;tpw=4;tpw=6;tpw=2,sweep=0" --envs 1 --rounds 7 --steps 40 --burnin 600 --pool 8 > gpurun_out/r4_tpw_n$N.log 2>&1
  echo N=$N; grep POOLED gpurun_out/r4_tpw_n$N.log | cut -c1-110
done
