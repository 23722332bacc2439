set -e
L=cur=rl_rocket_6dof_amd/csrc/libr6dof.so
timeout -k 10 400 python scripts/ab_swap.py --n 8388608 --libs $L --shapes "tpw=0;tpw=2;tpw=4" --envs 2 --rounds 7 --steps 40 --burnin 600 --pool 8 > gpurun_out/r4_tpw3_8m.log 2>&1
grep POOLED gpurun_out/r4_tpw3_8m.log | cut -c1-120; grep "^env" gpurun_out/r4_tpw3_8m.log | cut -c1-100
python bench.py --envs-per-gpu 8388608 --pool 8 --steps 200 --warmup 40 --burnin 600 --no-cpu-baseline --no-extras | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 2^23', l['ms_per_step'], l['roofline']['frac'])"
python scripts/clock_probe.py rl_rocket_6dof_amd/csrc/variants/libr6dof_clock.so 2>&1 | tail -8
python scripts/clock_probe.py rl_rocket_6dof_amd/csrc/variants/libr6dof_clockskel.so 2>&1 | tail -8
