"""Worker of tests/test_gpu_ppo.py::test_multi_rank_update_runs_from_a_graph_with_the_all_reduce_inside: ONE rank, a real RCCL
process group (backend nccl), R6_FORCE_PROCESS_GROUP=1 so that ppo.PPO takes its multi-rank code path -- the gradient all-reduce of
every minibatch step -- and captures it into the epoch graph.  A second trainer without collectives, same seeds, is the reference: an
all-reduce over one rank is the identity, so the weights must agree bit for bit.  Prints one JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29591")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
out = {}
flats = {}
for mode in ("collectives", "plain"):
    if mode == "collectives":
        os.environ["R6_FORCE_PROCESS_GROUP"] = "1"
    else:
        os.environ.pop("R6_FORCE_PROCESS_GROUP", None)
    env = BatchedRocket6DOF(4096, default_env_config(), remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, seed=1)
    algo = PPO(env, PPOConfig(n_steps=16, batch_size=1 << 13, n_epochs=3, net_arch=(64, 64), shared_trunk=False, seed=5))
    assert algo._collectives == (mode == "collectives")
    for _ in range(3):
        algo.collect_rollouts()
        info = algo.train()
    torch.cuda.synchronize()
    flats[mode] = torch.cat([p.detach().reshape(-1) for p in algo.policy.parameters()]).cpu()
    out[mode] = {"graph_captured": algo._graph is not None, "graph_collective_error": algo.graph_collective_error,
                 "fused_update": algo.fused_update is not None, "info": info}
    del algo, env
out["weights_identical"] = bool(torch.equal(flats["collectives"], flats["plain"]))
out["max_abs_diff"] = float((flats["collectives"] - flats["plain"]).abs().max())
out["backend"] = dist.get_backend()
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("RESULT " + json.dumps(out), flush=True)
