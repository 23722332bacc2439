set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
for L in rl_rocket_6dof_amd/csrc/libr6dof.so $V/libr6dof_dyn1.so $V/libr6dof_dyn2.so $V/libr6dof_dyn3.so $V/libr6dof_dyn4.so rl_rocket_6dof_amd/csrc/libr6dof.so; do
echo $L
timeout -k 10 300 python scripts/kstep_bench.py --n 65536 --k 16 64 --reps 9 --lib $L 2>/dev/null | grep "K="
done
