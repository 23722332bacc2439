# ppo_configs4 at two minibatch sizes (bench.py's side figure, called directly)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python - > gpurun_out/r5_ppo_cfg4_mb.log 2>&1 <<'P'
import json, torch, bench
print(json.dumps(bench.ppo_configs4_side_measurement(torch, torch.device("cuda:0")), indent=1))
P
cat gpurun_out/r5_ppo_cfg4_mb.log
