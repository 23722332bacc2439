#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched Rocket6DOF.step() hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 1 048 576
environments PER GPU (weak scaling: configs[3] = 8 x 2^20 sharded, no collective in step), fixed-step
RK4, fp32, config.yaml parameters (dt = 0.1 s), i.i.d. U(-1,1) actions pre-generated on the device
(two buffers used alternately, like a policy's output buffer; generation is outside the timed region), in-kernel auto-reset ON
so every env is live on every step.  A "step" is one r6_step launch over the rank's shard.  Inputs
are resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch / mean launch
duration measured with HIP events on the launch stream; algorithmic bytes = 189 B/env-step for the raw
env (read state 56 + m0 4 + action 12; write state 56 + obs 56 + reward 4 + done 1).
`cpu_baseline` = the reference-equivalent CPU restatement (oracle/, scalar Python: adaptive RK45 +
ground event + np.roots reward, the same algorithm and structure as the reference's Rocket6DOF.step)
timed on this box, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_RAW = 189
BYTES_WRAPPED = 201            # obs 13 floats (-4 B), ep_step r+w (+8), ep_return r+w (+8)


def cpu_baseline(seconds: float):
    """Times the oracle's reference-equivalent single-env loop on one host core."""
    from oracle import oracle_np as onp
    import numpy as np
    from rl_rocket_6dof_amd import default_env_config

    env = onp.OracleEnv(default_env_config(), integrator="adaptive", seed=42)
    rng = np.random.RandomState(0)
    env.reset()
    for _ in range(20):   # warm
        _, _, d, _ = env.step(rng.uniform(-1, 1, 3).astype(np.float32))
        if d:
            env.reset()
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            _, _, d, _ = env.step(rng.uniform(-1, 1, 3).astype(np.float32))
            steps += 1
            if d:
                env.reset()
    el = time.perf_counter() - t0
    out = {"value": steps / el, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{steps} random-action steps of one env (config.yaml, dt 0.1, adaptive RK45 + event + "
                     f"np.roots reward, reset on done) in {el:.1f} s on 1 of {os.cpu_count()} host cores"}
    c = cpu_baseline_c(max(2.0, seconds / 3))
    if c is not None:
        out["best_cpu"] = c
    return out


def cpu_baseline_c(seconds: float):
    """Optional 'best CPU' line: the oracle's C restatement (same fixed-step math as the HIP kernel,
    float64, OpenMP over envs) on all host cores.  None if the C oracle is not built."""
    try:
        from oracle import oracle_c
    except Exception:
        return None
    try:
        return oracle_c.benchmark(seconds)
    except Exception as e:  # pragma: no cover
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--integrator", default="rk4", choices=["rk4", "euler"])
    ap.add_argument("--dt", type=float, default=None, help="env timestep (default: config.yaml 0.1)")
    ap.add_argument("--variant", default="raw", choices=["raw", "wrapped"],
                    help="raw = Rocket6DOF.step (189 B); wrapped = fused make_env() stack (201 B)")
    ap.add_argument("--pool", type=int, default=2,
                    help="number of pre-generated action buffers cycled through.  In use the action tensor is what the "
                         "policy wrote a moment ago, i.e. resident in the 256 MB Infinity Cache when the step reads it; "
                         "a large pool of cold buffers adds traffic that does not exist then (1-2 buffers: 38.6 us per "
                         "step, 8: 40.5, 16: 44.3)")
    ap.add_argument("--burnin", type=int, default=2000,
                    help="untimed auto-reset steps run once after reset(), before warmup: all envs start in lockstep, "
                         "so without it the timed region would sit on the first mass-termination wave instead of the "
                         "steady mix of flight phases")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
    from rl_rocket_6dof_amd import dist as r6dist

    # R6_DIST_BACKEND=gloo + R6_FORCE_DEVICE=0 rehearse the N>1 code path with several ranks on ONE GPU
    # (RCCL refuses two ranks per device); the driver's real runs use the defaults (nccl, LOCAL_RANK).
    rank, world, local = r6dist.init_from_env(backend=os.environ.get("R6_DIST_BACKEND"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product has no CPU path)")
    local = int(os.environ.get("R6_FORCE_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    on_gpu_pg = world > 1 and dist.get_backend() == "nccl"

    n = args.envs_per_gpu
    total = n * world
    cfg = default_env_config()
    wrapped = args.variant == "wrapped"
    offset = rank * n
    env = BatchedRocket6DOF(
        n, cfg, device=dev, integrator=args.integrator, timestep=args.dt, env_offset=offset, auto_reset=True,
        record_flags=False, record_terms=False, track_episodes=wrapped,
        remove_mass=wrapped, clip_reward=(-1, 100) if wrapped else None,
        max_episode_steps=int(150 / (args.dt or cfg["timestep"])) if wrapped else 0, seed=42)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    pool = [(torch.rand((n, 3), generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(args.pool)]
    env.reset()
    for k in range(args.burnin):
        env.step_async(pool[k % args.pool])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        env.step_async(pool[k % args.pool])
    barrier()

    stream = torch.cuda.current_stream(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for k in range(args.steps):
        env.step_async(pool[k % args.pool])
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps      # mean launch-to-launch duration on the stream

    if world > 1:
        t = torch.tensor([wall, kernel_ms], dtype=torch.float64, device=dev if on_gpu_pg else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kernel_ms = float(t[0]), float(t[1])
        # the one collective of the design (episode summaries), exercised outside the timed region
        if env.stats is not None:
            r6dist.gather_episode_stats(env.episode_stats(), device=dev)

    if rank == 0:
        bytes_per = BYTES_WRAPPED if wrapped else BYTES_RAW
        achieved = bytes_per * n / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if os.path.exists(tpath) and not wrapped and args.integrator == "rk4" and n == (1 << 20):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        dt = args.dt or cfg["timestep"]
        line = {
            "metric": "env-steps/sec at N=2^20 envs per MI355X",
            "value": total * args.steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{n} envs/GPU x {world} GPU, {args.integrator} dt={dt} fp32, {args.variant} "
                                   f"Rocket6DOF.step, random actions, auto-reset, {args.burnin} burn-in steps",
                       "envs_total": total, "integrator": args.integrator, "dt": dt, "variant": args.variant,
                       "bytes_per_env_step": bytes_per, "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_us": kernel_ms * 1e3, "kernel": "r6::step_kernel"},
        }
        if world == 1 and not args.no_cpu_baseline and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
