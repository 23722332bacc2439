set -e
cd $GRAFT_REPO_ROOT
python3 bench.py --k 16 --envs-per-gpu 65536 --integrator euler --dt 0.04 --steps 1600 --warmup 160 --no-extras --no-cpu-baseline --preroll-seconds 0.3 > gpurun_out/r5_bench_k16.json 2> gpurun_out/r5_bench_k16.err || { tail -5 gpurun_out/r5_bench_k16.err; exit 1; }
python3 -c "
import json; b=json.load(open('gpurun_out/r5_bench_k16.json')); print('k16 line:', '%.3g' % b['value'], b['ms_per_step']*1e3, b['steady']['median'], b['config']['workload'][:120])"
python3 bench.py --variant wrapped --steps 400 --warmup 40 --no-extras --no-cpu-baseline --preroll-seconds 0.3 > gpurun_out/r5_bench_wrapped.json 2> gpurun_out/r5_bench_wrapped.err || { tail -5 gpurun_out/r5_bench_wrapped.err; exit 1; }
python3 -c "
import json; b=json.load(open('gpurun_out/r5_bench_wrapped.json')); print('wrapped line:', '%.3g' % b['value'], b['ms_per_step']*1e3, b['roofline']['frac'], b['steady']['median'])"
R6_FORCE_PROCESS_GROUP=1 python3 bench.py --steps 200 --warmup 20 --preroll-seconds 0.3 > gpurun_out/r5_bench_forcepg.json 2> gpurun_out/r5_bench_forcepg.err || { tail -5 gpurun_out/r5_bench_forcepg.err; exit 1; }
python3 -c "
import json; b=json.load(open('gpurun_out/r5_bench_forcepg.json')); print('1-rank RCCL line:', '%.3g' % b['value'], b['n_gpus'], b['rccl']['backend'], b['rccl']['with_rollout_gather']['us_per_gather'], 'steady' in b)"
