set -e
python -m pytest tests/test_gpu_bench_launcher.py tests/test_gpu_launch_shapes.py tests/test_gpu_fullsize.py tests/test_gpu_step_k.py -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r4_bench_final1.json 2> gpurun_out/r4_bench_final1.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4_bench_final1.json').read().strip().splitlines()[-1])
print('ms_per_step',l['ms_per_step'],'frac',l['roofline']['frac'],'kernel_us',l['roofline']['kernel_us'])
print('ooc',l['out_of_cache']['frac'],l['out_of_cache']['us_per_env_step'])
print('k16',l['configs1']['k16']['value'],l['configs1']['k16']['roofline'].get('frac'))
print('rk45',l['rk45']['us_per_env_step'],l['rk45']['frac'])
print('valu',l['valu']['issue_time_us_per_simd'],l['valu']['share_of_launch'])
print('ppo',l['ppo_training'].get('env_steps_per_s'),l['ppo_training'].get('minibatch_step_us'))
print('cpu',l['cpu_baseline']['value'],l['cpu_baseline']['all_cores']['value'])
PY
