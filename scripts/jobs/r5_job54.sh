# two ranks on the one GPU (gloo, R6_FORCE_DEVICE=0): the launcher test, then the DRIVER's shape (--steps 20 --warmup 5) with two ranks --
# each rank's clock now stops at its own synchronize, in front of the closing barrier: wall_s_max against steps x kernel time
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bench_launcher.py -x -q -m gpu -k "two_ranks or gpus_2" > gpurun_out/r5_two_rank_test.log 2>&1 || { tail -30 gpurun_out/r5_two_rank_test.log; exit 1; }
tail -1 gpurun_out/r5_two_rank_test.log
export R6_DIST_BACKEND=gloo R6_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --gpus 2 --envs-per-gpu 262144 --steps 20 --warmup 5 > gpurun_out/r5_bench_2rank_steps20_$i.json 2> gpurun_out/r5_bench_2rank_steps20_$i.err
  python3 -c "
import json
b=[json.loads(l) for l in open('gpurun_out/r5_bench_2rank_steps20_$i.json') if l.startswith('{')][-1]
r=b['rccl']; print('2 ranks x 262144 envs, 20 steps: ms_per_step', round(b['ms_per_step']*1e3,2), 'us; kernel_us', round(r['kernel_us_min'],2), round(r['kernel_us_max'],2), 'wall_s', r['wall_s_min'], r['wall_s_max'])"
done
