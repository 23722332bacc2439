# round 5, job 5: attempts per lane / tile of RK45 in the bench workload; env groups test + two-group figure; RCCL-in-graph rehearsal; driver-shape test
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 scripts/rk45_attempts_probe.py > gpurun_out/r5_rk45_attempts.json 2> gpurun_out/r5_rk45_attempts.err
python3 -c "
import json; d=json.load(open('gpurun_out/r5_rk45_attempts.json'))['rep2']
for k,v in d.items(): print(k, v)"
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "groups" > gpurun_out/r5_groups_test.log 2>&1 || { tail -30 gpurun_out/r5_groups_test.log; exit 1; }
tail -2 gpurun_out/r5_groups_test.log
timeout -k 10 600 python -m pytest tests/test_gpu_ppo.py -x -q -s -k "all_reduce_inside" > gpurun_out/r5_rccl_graph_test.log 2>&1 || { tail -40 gpurun_out/r5_rccl_graph_test.log; exit 1; }
grep "ppo rccl graph" gpurun_out/r5_rccl_graph_test.log; tail -2 gpurun_out/r5_rccl_graph_test.log
timeout -k 10 900 python -m pytest tests/test_gpu_bench_launcher.py -x -q -s -k "driver_shape" > gpurun_out/r5_driver_shape_test.log 2>&1 || { tail -40 gpurun_out/r5_driver_shape_test.log; exit 1; }
grep "driver shape" gpurun_out/r5_driver_shape_test.log; tail -2 gpurun_out/r5_driver_shape_test.log
python3 - <<'PY'
import json, sys, torch
sys.path.insert(0, ".")
import bench
from rl_rocket_6dof_amd import default_env_config
r = bench.two_groups_side_measurement(torch, torch.device("cuda:0"), default_env_config())
json.dump(r, open("gpurun_out/r5_two_groups.json", "w"), indent=1)
print({k: v for k, v in r.items() if k != "workload"})
PY
