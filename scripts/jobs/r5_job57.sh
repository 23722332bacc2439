# after the bench's multi-rank section was made independent of the device collectives for its record: two gloo ranks on the one GPU
# (launcher test), one rank on a real RCCL group (R6_FORCE_PROCESS_GROUP=1: host gloo group for the times, RCCL for the rollout gathers)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bench_launcher.py -x -q -m gpu -k "gpus_2" > gpurun_out/r5_two_rank_test.log 2>&1 || { tail -30 gpurun_out/r5_two_rank_test.log; exit 1; }
tail -1 gpurun_out/r5_two_rank_test.log
R6_FORCE_PROCESS_GROUP=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_forced_pg.json 2> gpurun_out/r5_bench_forced_pg.err || { tail -20 gpurun_out/r5_bench_forced_pg.err; exit 1; }
echo "stdout lines: $(wc -l < gpurun_out/r5_bench_forced_pg.json)"
python3 -c "
import json
b=json.load(open('gpurun_out/r5_bench_forced_pg.json')); r=b['rccl']
print('forced one-rank RCCL group: ms_per_step', round(b['ms_per_step']*1e3,2), 'kernel', round(b['roofline']['kernel_us'],2), 'backend', r['backend'], 'gather', {k: r['with_rollout_gather'].get(k) for k in ('gathers','episodes_seen','us_per_gather','error')})"
