# final tree: the whole GPU suite, smoke, the default bench command and the driver's command shape
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputest_final2.log 2>&1 || { tail -40 gpurun_out/r5_gputest_final2.log; exit 1; }
tail -1 gpurun_out/r5_gputest_final2.log
python -c "import __graft_entry__ as g; g.smoke()"
python3 bench.py > gpurun_out/r5_bench_default2.json 2> gpurun_out/r5_bench_default2.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_shape2.json 2> gpurun_out/r5_bench_driver_shape2.err
python3 -c "
import json
for f in ('r5_bench_default2', 'r5_bench_driver_shape2'):
    b = json.load(open('gpurun_out/%s.json' % f))
    print(f, 'wall', round(b['ms_per_step']*1e3,2), 'kernel', round(b['roofline']['kernel_us'],2), 'frac', round(b['roofline']['frac'],3), 'steady median', round(b['steady']['median'],2), 'lib', b['config']['lib_commit'])
    print('   configs0', {k: round(v) for k, v in b['configs0'].items() if k.endswith('_per_s')}, 'k16', '%.3g' % b['configs1']['k16']['value'], 'rk45', round(b['rk45']['us_per_env_step'],1), 'ppo', '%.3g' % b['ppo_training']['value'] if 'value' in b['ppo_training'] else '')
"
