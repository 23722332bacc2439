set -e
B="python bench.py --envs-per-gpu 8388608 --pool 8 --steps 200 --warmup 40 --burnin 600 --no-extras --no-cpu-baseline"
for rep in 1 2 3; do
  for T in 2 4 3; do
    R6_BALANCED_TPW=$T $B | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tpw $T rep $rep', round(l['ms_per_step']*1e3,1), round(l['roofline']['frac'],3))"
  done
done
