set -e
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 600 python -m pytest tests/test_gpu_step_k.py -m gpu -x -q 2>&1 | tail -3
for L in $V/libr6dof_pairc0.so rl_rocket_6dof_amd/csrc/libr6dof.so; do
  echo $L
  timeout -k 10 300 python scripts/kstep_bench.py --n 65536 --k 4 16 64 --lib $L 2>&1 | tail -3
  timeout -k 10 300 python scripts/kstep_bench.py --n 65536 --k 16 --integrator rk4 --dt 0.1 --lib $L 2>&1 | tail -1
  timeout -k 10 300 python scripts/kstep_bench.py --n 16384 --k 16 --lib $L 2>&1 | tail -1
done
timeout -k 10 300 python scripts/stepk_soak.py 10000 2>&1 | tail -2
