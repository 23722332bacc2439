"""GPU-free stand-in for one RANK of a self-launched ``bench.py --gpus N`` job (tests/test_bench_line.py): everything bench.main()
does around its timed loops -- argument and WORLD_SIZE checks, process group from torchrun's environment (gloo), reduction of the
per-rank times, the `rccl` object, rank 0's ONE JSON line -- with the GPU part replaced by fixed per-rank numbers.  Test
infrastructure: bench.py never imports it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    args = bench.parse_args(sys.argv[1:])
    world = bench.check_world(args)
    assert world is not None and "R6_BENCH_LAUNCHER_PID" in os.environ, "the stub is a rank: it must run under the launcher"
    if os.environ.get("R6_STUB_FAIL_RANK") == os.environ.get("RANK"):
        raise SystemExit(7)
    if os.environ.get("R6_STUB_PIDDIR"):        # "a rank that runs for a long time": leaves its pid and sleeps
        import time

        open(os.path.join(os.environ["R6_STUB_PIDDIR"], f"rank{os.environ.get('RANK')}.pid"), "w").write(str(os.getpid()))
        time.sleep(float(os.environ.get("R6_STUB_SLEEP", "120")))
    import torch.distributed as dist

    from rl_rocket_6dof_amd import dist as r6dist

    rank, w, _ = r6dist.init_from_env(backend="gloo")
    wall, kernel_ms = (0.080, 0.0380) if rank == 0 else (0.084, 0.0402)
    red = r6dist.reduce_bench_times(wall, kernel_ms)
    tot, _ = r6dist.gather_episode_stats({k: float(rank + 1) for k in r6dist.STAT_NAMES})
    extras = {"rccl": {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "devices": red["devices"],
                       "with_rollout_gather": {"episodes_seen": tot["episodes"]}}}
    line = bench.build_line(args, w, args.envs_per_gpu, 0.1, red["wall_max"], red["kernel_ms_max"], extras=extras)
    line["data"] = "stub"                       # nobody can mistake a stand-in's line for a measurement
    if os.environ.get("R6_STUB_WRONG_WORLD") == "1":
        line["n_gpus"] = 1
    if rank == 0:
        print(json.dumps(line), flush=True)
    if os.environ.get("R6_STUB_FAIL_AFTER_LINE") == os.environ.get("RANK"):     # "a rank that crashes in the teardown"
        dist.barrier()
        raise SystemExit(9)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
