#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched Rocket6DOF.step() hot path.

    python bench.py --gpus N --steps K --warmup W          # N > 1 from a plain shell: starts its own N ranks (launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 1 048 576
environments PER GPU (weak scaling: configs[3] = 8 x 2^20 sharded, no collective in step), fixed-step
RK4, fp32, config.yaml parameters (dt = 0.1 s), i.i.d. U(-1,1) actions pre-generated on the device
(two buffers used alternately, like a policy's output buffer; generation is outside the timed region), in-kernel auto-reset ON
so every env is live on every step.  A "step" is one r6_step launch over the rank's shard (with --k K > 1: one of
the K env steps of an r6_step_k launch).  Inputs are resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch / mean launch
duration measured with HIP events on the launch stream; algorithmic bytes = 189 B/env-step for the raw
env (read state 56 + m0 4 + action 12; write state 56 + obs 56 + reward 4 + done 1).
Beside the headline the same line carries (N = 1 only, all timed in this run):
  `out_of_cache`  the same kernel on 2^23 envs with 8 cold action buffers: a 1.6 GB working set, far beyond the 256 MB
                  Infinity Cache that holds the whole headline working set (198 MB);
  `configs1`      BASELINE configs[1] -- 65 536 envs, Euler dt 0.04 -- one launch per step and 16 steps per launch
                  (r6_step_k), each with its own algorithmic-bytes figure;
  `rk45`          the headline size stepped by the in-kernel adaptive RK45 twin of the reference's integrator;
  `cpu_baseline`  the reference-equivalent CPU restatement (oracle/, scalar Python: adaptive RK45 + ground event +
                  np.roots reward, the same algorithm and structure as the reference's Rocket6DOF.step) on this box.
With N > 1 the line carries `rccl`: the process-group facts (backend, ranks), per-rank kernel times, and a second short
timed loop that includes the design's one collective -- the all-gather of episode summaries -- every 32 steps
(BASELINE configs[4]'s per-rollout exchange).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_RAW = 189
BYTES_WRAPPED = 201            # obs 13 floats (-4 B), ep_step r+w (+8), ep_return r+w (+8)
TRAFFIC_PROFILE = os.path.join("profiles", "r5_pmc_traffic.json")


def bytes_per_env_step(variant: str, k: int = 1, obs_dim: int = 14) -> float:
    """Algorithmic bytes per env-step.  One launch per step: SURVEY 8(d).  K steps per launch (r6_step_k): the action
    read and the per-step outputs (obs, reward, done) stay, the state / m0 / v0 (and episode counter) traffic is paid
    once per K steps."""
    if k <= 1:
        return float(BYTES_WRAPPED if variant == "wrapped" else BYTES_RAW)
    per_step = 12 + 4 * obs_dim + 4 + 1
    once = 56 + 56 + 2 * 8 + (16 if variant == "wrapped" else 0)      # state r+w, epconst rows r+w, ep_step / ep_return r+w
    return per_step + once / k


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
    out = {"value": steps / el, "unit": "env-steps/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
           "sample": f"{steps} random-action steps of one env (config.yaml, dt 0.1, adaptive RK45 + event + "
                     f"np.roots reward, reset on done) in {el:.1f} s on 1 of {os.cpu_count()} host cores"}
    out["all_cores"] = cpu_baseline_all_cores(max(3.0, seconds / 2))
    c = cpu_baseline_c(max(2.0, seconds / 3))
    if c is not None:
        out["best_cpu"] = c
    return out


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores() -> int:
    """Cores this process may actually use: the scheduler affinity mask, cut by a cgroup CPU quota where one is set (a GPU
    box hands a one-GPU job a share of its host cores, not all of os.cpu_count())."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    return max(1, n)


_WORKER = r"""
import json, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from oracle import oracle_np as onp
cfg, seconds, seed = json.loads(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
env = onp.OracleEnv(cfg, integrator="adaptive", seed=seed)
rng = np.random.RandomState(seed)
env.reset()
for _ in range(20):
    _, _, d, _ = env.step(rng.uniform(-1, 1, 3).astype(np.float32))
    if d:
        env.reset()
print("ready", flush=True)
sys.stdin.readline()                      # all workers start their timed region together
steps, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < seconds:
    for _ in range(50):
        _, _, d, _ = env.step(rng.uniform(-1, 1, 3).astype(np.float32))
        steps += 1
        if d:
            env.reset()
print(json.dumps({"steps": steps, "seconds": time.perf_counter() - t0}), flush=True)
"""


def cpu_baseline_all_cores(seconds: float, max_procs: int = 64):
    """the same reference-equivalent loop as N independent processes, one per usable host core (the
    reference is single-env, single-process Python: running one copy per core is how it scales on a host).  The workers are
    fresh interpreters that import numpy / scipy and the oracle only -- they never touch the GPU."""
    from rl_rocket_6dof_amd import default_env_config

    cores = usable_cores()
    procs_n = min(cores, max_procs)
    cfg = json.dumps(default_env_config())
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = []
    try:
        for i in range(procs_n):
            procs.append(subprocess.Popen([sys.executable, "-c", _WORKER, ROOT, cfg, str(seconds), str(1000 + i)], env=env,
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("worker did not start")
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
        for p in procs:
            p.wait(timeout=30)
    except Exception as e:  # pragma: no cover
        for p in procs:
            if p.poll() is None:
                p.kill()
        return {"error": repr(e), "cores": procs_n}
    total = sum(r["steps"] for r in res)
    el = max(r["seconds"] for r in res)
    return {"value": total / el, "unit": "env-steps/s", "cores": procs_n, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{procs_n} independent processes (one per usable core: {cores} usable of {os.cpu_count()} on the host), each the "
                      f"single-env loop above for {seconds:.1f} s: {total} steps in all"}


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--integrator", default="rk4", choices=["rk4", "euler"])
    ap.add_argument("--dt", type=float, default=None, help="env timestep (default: config.yaml 0.1)")
    ap.add_argument("--variant", default="raw", choices=["raw", "wrapped"],
                    help="raw = Rocket6DOF.step (189 B); wrapped = fused make_env() stack (201 B)")
    ap.add_argument("--k", type=int, default=1,
                    help="env steps per kernel launch (r6_step_k; 1 = one r6_step launch per step, the headline). --steps "
                         "still counts env steps and must be a multiple of k")
    ap.add_argument("--pool", type=int, default=2,
                    help="number of pre-generated action buffers cycled through.  In use the action tensor is what the "
                         "policy wrote a moment ago, i.e. resident in the 256 MB Infinity Cache when the step reads it; "
                         "a large pool of cold buffers adds traffic that does not exist then (1-2 buffers: 38.6 us per "
                         "step, 8: 40.5, 16: 44.3)")
    ap.add_argument("--burnin", type=int, default=2000,
                    help="untimed auto-reset steps run once after reset(), before warmup: all envs start in lockstep, "
                         "so without it the timed region would sit on the first mass-termination wave instead of the "
                         "steady mix of flight phases")
    ap.add_argument("--preroll-seconds", type=float, default=1.0,
                    help="untimed stepping by TIME (not by count) after the burn-in and before --warmup: the chip's clocks and the "
                         "launch pipeline are in their steady state when the contract's W + K steps run (2000 burn-in launches are "
                         "75 ms; the first 30 ms of a process run 32 -> 53 -> 36 us per launch, gpurun_out/r5_trace_driver_shape_summary.json)")
    ap.add_argument("--preroll-batch", type=int, default=4,
                    help="launches between two synchronizes of the pre-roll (4 = 92 %% duty: the timed region does not start at the chip's "
                         "power limit; 200 = round 5's first form, 99 %% duty: see preroll())")
    ap.add_argument("--steady-launches", type=int, default=2000,
                    help="launches of the `steady` record taken right AFTER the contract's timed region on the same env, in chunks of "
                         "--steady-chunk launches (0 = skip): the distribution a K-launch sample is drawn from")
    ap.add_argument("--steady-chunk", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the out_of_cache / configs1 side measurements (N = 1)")
    ap.add_argument("--gather-every", type=int, default=32,
                    help="N > 1: period (steps) of the episode-summary all-gather in the second timed loop")
    return ap.parse_args(argv)


def lib_commit() -> str:
    """Source commit the HIP LIBRARY was built from: the stamp the build left next to libr6dof.so (it travels with the .so to
    machines without .git).  bench.py / dist.py may be newer than it: see bench_py_sha16."""
    try:
        return open(os.path.join(ROOT, "rl_rocket_6dof_amd", "csrc", "BUILD_COMMIT")).read().strip() or "unknown"
    except Exception:
        return "unknown"


def bench_py_sha16() -> str:
    """First 16 hex digits of sha256(bench.py): identifies THIS file's version where there is no .git (the driver records the
    same digest beside its run)."""
    import hashlib

    try:
        return hashlib.sha256(open(os.path.abspath(__file__), "rb").read()).hexdigest()[:16]
    except Exception:
        return "unknown"


def traffic_from_profile(variant: str, integrator: str, n: int, k: int):
    """HBM bytes per launch from the committed PMC profile of THIS workload (separate --pmc passes, scripts/
    profile_round.sh).  It is a file constant, not measured in this run: `traffic_source` says so."""
    path = os.path.join(ROOT, TRAFFIC_PROFILE)
    if not (os.path.exists(path) and variant == "raw" and integrator == "rk4" and n == (1 << 20) and k == 1):
        return None, None
    try:
        prof = json.load(open(path))
        return prof.get("hbm_bytes_per_launch"), f"{TRAFFIC_PROFILE} (static file, PMC passes at commit {prof.get('commit', 'unrecorded')})"
    except Exception:
        return None, None


VALU_PROFILE = os.path.join("profiles", "r5_pmc_valu.json")
SIMDS = 1024                               # 256 CUs x 4 SIMDs
# Measured issue cost of a wave64 vector instruction per SIMD once two or more waves are ready (scripts/valu_operand_bench.hip,
# profiles/r3_experiments.json): vector / constant sources 1.2 ns, a scalar-register (SGPR / vcc) source 1.8 ns, a transcendental
# 3.4 ns.  (SQ_ACTIVE_INST_VALU counts a quad-cycle per instruction; pricing an instruction at those 4 cycles, as round 3's line
# did, overstates the pipe's occupancy by 1.5-2x.)
VALU_NS_PLAIN, VALU_NS_SCALAR_SRC, VALU_NS_TRANS = 1.2, 1.8, 3.4
SCALAR_SRC_SHARE_HOT_LOOP = 0.15           # static share of the tile loop's vector instructions with an SGPR / vcc source (scripts/valu_mix.py)


def valu_issue_time_us(insts_per_launch: float, trans_per_launch: float, simds: int = SIMDS) -> dict:
    """Per-SIMD VALU issue time of a launch at the measured rates: a floor (every instruction at the plain rate) and the priced
    figure (transcendentals and the scalar-source share at their own rates)."""
    lo = insts_per_launch * VALU_NS_PLAIN / simds * 1e-3
    hi = (insts_per_launch * VALU_NS_PLAIN + trans_per_launch * (VALU_NS_TRANS - VALU_NS_PLAIN) +
          SCALAR_SRC_SHARE_HOT_LOOP * insts_per_launch * (VALU_NS_SCALAR_SRC - VALU_NS_PLAIN)) / simds * 1e-3
    return {"floor_us": lo, "priced_us": hi}


def valu_from_profile(variant: str, integrator: str, n: int, k: int, kernel_ms: float):
    """The kernel's OTHER resource next to HBM: the time one SIMD needs to ISSUE the launch's vector instructions, from the
    instruction counts of the committed PMC pass of THIS workload (SQ_INSTS_VALU, SQ_INSTS_VALU_TRANS_F32; scripts/pmc.sh) priced at
    the measured issue rates, as a share of this run's kernel time.  A file constant like `traffic`."""
    path = os.path.join(ROOT, VALU_PROFILE)
    if not (os.path.exists(path) and variant == "raw" and integrator == "rk4" and n == (1 << 20) and k == 1 and kernel_ms > 0):
        return None
    try:
        prof = json.load(open(path))
        raw = prof["raw_per_launch"]
        insts = float(raw["SQ_INSTS_VALU"])
        trans = float(raw.get("SQ_INSTS_VALU_TRANS_F32", 0.052 * insts))
        t = valu_issue_time_us(insts, trans)
        return {"issue_time_us_per_simd": t["priced_us"], "issue_time_floor_us_per_simd": t["floor_us"],
                "share_of_launch": t["priced_us"] / (kernel_ms * 1e3),
                "instructions_per_64_env_tile": prof["derived"]["valu_instructions_per_64_env_tile"],
                "mean_resident_waves_per_simd": prof["derived"].get("mean_resident_waves_per_simd (SQ_WAVE_CYCLES * 4 / 1024 / kernel cycles)"),
                "pricing_ns_per_instruction": {"vector_or_constant_sources": VALU_NS_PLAIN, "scalar_register_source": VALU_NS_SCALAR_SRC,
                                               "transcendental": VALU_NS_TRANS, "scalar_source_share_assumed": SCALAR_SRC_SHARE_HOT_LOOP},
                "source": f"{VALU_PROFILE} (static file, PMC passes at commit {prof.get('commit', 'unrecorded')}); rates: scripts/valu_operand_bench.hip",
                "note": "the launch is bounded by how its memory time (30.6 us for the same accesses without arithmetic) and this issue "
                        "time overlap, not by either alone: ~3 of the 5 resident waves per SIMD are alive on average"}
    except Exception:
        return None


STEPK_PROFILE = os.path.join("profiles", "r5_pmc_stepk.json")


def small_batch_bound(n: int, integrator: str, dt, k: int, launch_s: float) -> dict:
    """BASELINE configs[1] (65 536 envs = 1024 tiles on 1024 SIMDs): a launch is bounded by the DEPENDENT chain of one tile's K
    steps on its SIMD (r6::step_k_pair_kernel: a dynamics wave, two reward waves and an output wave per tile), not by HBM and not by VALU issue:
    the committed PMC pass of this workload (scripts/pmc_stepk.sh) has the waves waiting (s_waitcnt / sleep: a partner's result, a
    load) for about half of their cycles.  Reported: the share of the launch one SIMD needs to issue the tile's vector instructions
    at the measured rates (see valu_issue_time_us) and the profile's split of the wave cycles."""
    if k <= 1:
        return {"bound": "launch_latency", "launch_us": launch_s * 1e6,
                "note": "one r6_step launch per env step: launch-to-launch time of a kernel whose tiles (one per wave) take one pass; the "
                        "memory time of the step is ~1.5 us, its one-wave-per-SIMD compute ~5 us -- r6_step_k amortises the rest"}
    out = {"bound": "dependent_chain_of_one_tile_per_simd", "unit": "share of the launch in which a SIMD issues vector instructions (measured issue rates)",
           "peak": 1.0, "launch_us": launch_s * 1e6}
    path = os.path.join(ROOT, STEPK_PROFILE)
    if os.path.exists(path) and n == 65536 and integrator == "euler" and k == 16:
        try:
            prof = json.load(open(path))
            raw = prof["raw_per_launch"]
            t = valu_issue_time_us(float(raw["SQ_INSTS_VALU"]), float(raw.get("SQ_INSTS_VALU_TRANS_F32", 0.0)))
            out["achieved"] = t["priced_us"] / (launch_s * 1e6)
            out["frac"] = out["achieved"]
            out["valu_issue_time_us_per_simd"] = t["priced_us"]
            out["valu_instructions_per_tile_and_env_step"] = prof["derived"]["valu_instructions_per_tile_and_env_step"]
            out["kernel"] = prof.get("kernel")
            out["share_of_wave_cycles"] = prof["derived"]["share_of_wave_cycles"]
            out["source"] = f"{STEPK_PROFILE} (static file, PMC passes at commit {prof.get('commit', 'unrecorded')}); rates: scripts/valu_operand_bench.hip"
        except Exception:
            pass
    return out


def build_line(args, world: int, n: int, dt: float, wall: float, kernel_ms: float, *, traffic=None, traffic_source=None,
               extras=None) -> dict:
    """The ONE JSON line of the bench contract from the reduced measurements (pure function: testable without a GPU).
    wall = seconds of the timed region (max over ranks); kernel_ms = mean milliseconds per env step on the launch
    stream (max over ranks)."""
    total = n * world
    bytes_per = bytes_per_env_step(args.variant, args.k, 13 if args.variant == "wrapped" else 14)
    achieved = bytes_per * n / (kernel_ms * 1e-3) / 1e9
    kernel = "r6::step_kernel" if args.k <= 1 else f"r6::step_k_kernel (K={args.k})"
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
                               f"Rocket6DOF.step, random actions, auto-reset, {args.burnin} burn-in steps + {args.preroll_seconds:g} s of "
                               f"untimed stepping (a synchronize every {args.preroll_batch} launches) before the {args.warmup} warm-up and {args.steps} timed steps, "
                               f"{args.k} env step(s) per launch",
                   "envs_total": total, "integrator": args.integrator, "dt": dt, "variant": args.variant,
                   "steps_per_launch": args.k, "bytes_per_env_step": bytes_per, "parallelism": f"env-shard x{world}",
                   "lib_commit": lib_commit(), "bench_py_sha16": bench_py_sha16()},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_us": kernel_ms * 1e3 * max(args.k, 1), "kernel": kernel,
                     "working_set_note": (f"{bytes_per:g} B x {n} envs = {n * bytes_per / 1e6:.0f} MB per launch fits the 256 MB Infinity "
                                          "Cache: see out_of_cache for the same kernel against HBM proper") if n * bytes_per < 256e6 else None},
    }
    if extras:
        line.update(extras)
    return line


def timed_loop(env, pool, steps: int, k: int, barrier, stream, torch, every=None):
    """Times `steps` env steps (launches of k steps each) between two barriers (each = synchronize + process-group barrier +
    synchronize); returns (this rank's wall seconds from behind the opening barrier to its own synchronize in front of the closing
    one, ms per env step on the launch stream).  `every` = (period, fn): fn() is called after each `period` steps inside the timed region."""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    npool = len(pool)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    if k <= 1:
        for j in range(steps):
            env.step_async(pool[j % npool])
            if every is not None and (j + 1) % every[0] == 0:
                every[1]()
    else:
        for j in range(steps // k):
            env.step_k(pool[j % npool], out=env._bench_out)
            if every is not None and ((j + 1) * k) % every[0] == 0:
                every[1]()
    ev1.record(stream)
    torch.cuda.synchronize()           # THIS rank's `steps` steps are done: its clock stops here ...
    wall = time.perf_counter() - t0
    barrier()                          # ... and the closing barrier follows.  The job's time is the MAX over the ranks' clocks
    #                                    (reduce_bench_times), as the contract prescribes; with the barrier inside the clock every rank
    #                                    would read the same time plus the barrier's own latency -- a host-side gloo barrier of two
    #                                    ranks cost 140 us in round 4's two-rank record (9.857 ms wall against 9.716 ms of kernels),
    #                                    a fifth of the driver's 0.8 ms timed region, charged to nobody's env steps.  At N = 1 the
    #                                    barrier is a second synchronize and nothing changes.
    return wall, ev0.elapsed_time(ev1) / steps


def make_pool(torch, n, k, count, dev, seed):
    gen = torch.Generator(device=dev).manual_seed(seed)
    shape = (n, 3) if k <= 1 else (k, n, 3)
    return [(torch.rand(shape, generator=gen, device=dev) * 2 - 1).contiguous() for _ in range(count)]


def preroll(env, pool, seconds: float, k: int, sync, batch: int = 4) -> int:
    """Untimed stepping for `seconds` of wall time, in batches of `batch` launches with a synchronize between them; returns the number
    of env steps issued.  Why small batches (scripts/preroll_duty_probe.py): behind 1 s of stepping in batches of
    200 (99 % duty) the chip is at its power limit, and a 20-launch timed region started there sat in a clock dip in 3 of 12 trials
    (35.5 ... 43.1 us); behind 1 s in batches of 4 (92 % duty, 25 000 launches) it never did in 12 (35.4 ... 36.7 us = 0.97 ... 1.01 x the
    steady median measured right behind it), and behind one launch per synchronize (70 % duty) neither -- but there the dips arrive
    two to four chunks later.  The timed region itself is back to back, as the contract says; a 2000-step region meets the dips
    anyway, and the `steady` record shows them."""
    done, npool = 0, len(pool)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        if k <= 1:
            for j in range(batch):
                env.step_async(pool[j % npool])
            done += batch
        else:
            for j in range(max(1, batch // k)):
                env.step_k(pool[j % npool], out=env._bench_out)
            done += max(1, batch // k) * k
        sync()
    return done


def steady_record(env, pool, launches: int, chunk: int, k: int, stream, torch, sync) -> dict:
    """`launches` back-to-back launches on the same env right after the contract's timed region, HIP events between chunks of
    `chunk` launches: the distribution of CHUNK MEANS (us per env step) a `chunk`-launch timed region is a sample of.  Slow
    episodes (1.5-3 ms at +5 ... +15 %, arriving every 8-25 ms, not replaying with the data: DESIGN.md section 6) show as the
    distance between p90 / max and the median."""
    chunks = max(1, launches // chunk)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    npool = len(pool)
    per_launch = max(k, 1)
    sync()
    ev[0].record(stream)
    for c in range(chunks):
        for j in range(chunk):
            if k <= 1:
                env.step_async(pool[j % npool])
            else:
                env.step_k(pool[j % npool], out=env._bench_out)
        ev[c + 1].record(stream)
    sync()
    us = sorted(ev[c].elapsed_time(ev[c + 1]) / (chunk * per_launch) * 1e3 for c in range(chunks))
    q = lambda f: us[min(chunks - 1, int(f * chunks))]
    med = q(0.5)
    return {"launches": chunks * chunk, "chunk_launches": chunk, "chunks": chunks, "unit": "us per env step, HIP events, chunk means",
            "mean": sum(us) / chunks, "median": med, "p10": q(0.1), "p90": q(0.9), "min": us[0], "max": us[-1],
            "share_of_chunks_above_1.06_x_median": sum(1 for x in us if x > 1.06 * med) / chunks}


def side_measurement(torch, dev, cfg, *, n, integrator, dt, k, pool_count, burnin, steps, warmup, label):
    """One secondary workload timed in the same run (N = 1): returns a dict with its own algorithmic-bytes figure."""
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    env = BatchedRocket6DOF(n, cfg, device=dev, integrator=integrator, timestep=dt, auto_reset=True, record_flags=False,
                            record_terms=False, track_episodes=False, seed=42)
    pool = make_pool(torch, n, k, pool_count, dev, 4321)
    env.reset()
    if k <= 1:
        for j in range(burnin):
            env.step_async(pool[j % pool_count])
    else:
        for j in range(max(1, burnin // k)):
            env.step_k(pool[j % pool_count], last_only=True)
        env._bench_out = env.step_k(pool[0])
    sync = lambda: torch.cuda.synchronize(dev)
    stream = torch.cuda.current_stream(dev)
    timed_loop(env, pool, warmup, k, sync, stream, torch)
    wall, ms = timed_loop(env, pool, steps, k, sync, stream, torch)
    bpe = bytes_per_env_step("raw", k)
    ach = bpe * n / (ms * 1e-3) / 1e9
    out = {"workload": label, "envs": n, "integrator": integrator, "dt": dt or cfg["timestep"], "steps_per_launch": k,
           "steps": steps, "action_buffers": pool_count, "value": n * steps / wall, "unit": "env-steps/s",
           "us_per_env_step": ms * 1e3, "bytes_per_env_step": bpe, "achieved": ach, "working_set_MB": bpe * n / 1e6}
    if bpe * n < 32e6:
        # a working set that never leaves the 32 MB of L2 / the Infinity Cache is not priced against HBM: the bound that
        # applies to at most ~one 64-env tile per SIMD is how much of a launch the SIMDs spend issuing VALU work
        out["roofline"] = small_batch_bound(n, integrator, dt, k, ms * 1e-3 * k)
        out["hbm_frac_for_reference_only"] = ach / HBM_PEAK_GBS
    else:
        out["frac"] = ach / HBM_PEAK_GBS
    del env, pool
    torch.cuda.empty_cache()
    return out


def two_groups_side_measurement(torch, dev, cfg, rounds: int = 9, steps: int = 400) -> dict:
    """The headline workload stepped as TWO env groups of 2^19 envs, one r6_step launch each on two streams the env owns
    (BatchedRocket6DOF(groups=2, group_sync="none"): two independent pipelines, as an asynchronous vector env with one policy call
    per group drives them) against ONE launch per step -- on the same buffers, in alternating rounds of `steps` steps, wall clock
    between synchronisations.  A labelled side figure, never the line's `value`: the headline stays one launch per step."""
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    n = 1 << 20
    out = {"workload": "2^20 envs, RK4 dt 0.1, raw step, as two groups of 2^19 envs on two streams (no per-step fork / join) vs one launch per step; "
                       f"same buffers, {rounds} alternating rounds of {steps} steps, wall clock"}
    try:
        env = BatchedRocket6DOF(n, cfg, device=dev, groups=2, group_sync="none", auto_reset=True, record_flags=False, record_terms=False,
                                track_episodes=False, seed=42)
        pool = make_pool(torch, n, 1, 2, dev, 977)
        env.reset()
        torch.cuda.synchronize(dev)
        for j in range(2000):
            env.step_async(pool[j % 2])
        env.join_groups()
        res = {1: [], 2: []}
        for r in range(rounds):
            for g in ((1, 2) if r % 2 == 0 else (2, 1)):
                env.groups = g                    # the same env, the same memory: only the number of launches per step changes
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for j in range(steps):
                    env.step_async(pool[j % 2])
                env.join_groups()
                torch.cuda.synchronize(dev)
                res[g].append((time.perf_counter() - t0) / steps * 1e6)
        env.groups = 2
        med = lambda v: sorted(v)[len(v) // 2]
        one, two = med(res[1]), med(res[2])
        out.update({"one_launch_us_per_step": one, "two_groups_us_per_step": two, "gain_us": one - two,
                    "two_groups_env_steps_per_s": n / (two * 1e-6), "rounds_one": res[1], "rounds_two": res[2],
                    "frac_of_hbm_peak_two_groups": BYTES_RAW * n / (two * 1e-6) / 1e9 / HBM_PEAK_GBS})
        del env, pool
        torch.cuda.empty_cache()
    except Exception as e:   # noqa: BLE001 -- a side figure must not cost the line
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    return out


def single_env_side_measurement(torch, dev, steps: int = 250, repeats: int = 2) -> dict:
    """BASELINE configs[0]: ONE env stepped from the host, 1000 random-action steps with a reset at every episode end -- the
    reference's own way of running (test_6DOF_environment.py:23-35, test_6DOF_simulator.py:3-7), through the single-env drop-ins
    `Rocket6DOF` / `Simulator6DOF` (rocket_env.py of this package: one lane of one wave does the work; a step is one launch,
    two small copies through pinned images and one synchronisation).  Latency-bound by construction; a side figure.  Kept short on
    purpose (2 x 250 steps behind 100 per integrator; scripts/single_env_rate.py is the 5 x 1000-step form): the one-env RK4 launches
    are launches of the headline's kernel instance, and a `rocprofv3 --stats` of this command averages over all of them -- 600 launches
    of a few us next to the ~48 000 full-size ones move that average by 1 %."""
    import numpy as np

    from rl_rocket_6dof_amd import Rocket6DOF, Simulator6DOF, default_env_config
    rng = np.random.RandomState(0)

    def run(env, k):
        env.reset()
        t0 = time.perf_counter()
        for _ in range(k):
            if env.step(rng.uniform(-1, 1, 3).astype(np.float32))[2]:
                env.reset()
        return k / (time.perf_counter() - t0)

    out = {"label": f"BASELINE configs[0]: one env, host-driven, {steps} random-action steps, dt 0.1; env-steps/s (median of "
                    f"{repeats}); the reference's Python on one Xeon core of the survey container: 730/s (BASELINE.md section 2)",
           "steps": steps, "reference_python_steps_per_s": 730.0}
    for integ in ("rk4", "rk45"):
        env = Rocket6DOF(**default_env_config(), device=dev, integrator=integ)
        run(env, 100)
        out[f"rocket6dof_{integ}_steps_per_s"] = float(np.median([run(env, steps) for _ in range(repeats)]))
        env.close()
    ic = [100, 100, 100, 0, 3, 4, 1, 1, 0, 0, 0, 0, 0, 50e3]
    sim = Simulator6DOF(ic, dt=0.1, device=dev)

    def run_sim(k):
        t0 = time.perf_counter()
        for _ in range(k):
            sim.state = np.asarray(ic, dtype=np.float64)
            sim.step([0.0, 0.0, 1.0])
        del sim.states[1:], sim.actions[1:], sim.times[1:]
        return k / (time.perf_counter() - t0)

    run_sim(100)
    out["simulator6dof_rk4_steps_per_s"] = float(np.median([run_sim(steps) for _ in range(repeats)]))
    return out


def ppo_side_measurement(torch, dev, iters: int = 5, warm: int = 3) -> dict:
    """SURVEY section 8 (f2), the loop main_6DOF.py:107-139 drives through SB3: PPO iterations of the on-device trainer at the
    configuration of examples/train_ppo.py -- rollout + GAE, then 10 epochs x 16 minibatch steps -- timed whole (wall clock around
    synchronised iterations).  A side figure of the line, never its `value`; an exception is reported, not raised."""
    import time

    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
    from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

    n, steps = 16384, 64
    out = {"workload": "PPO on 16 384 wrapped envs (remove-mass obs, reward clip, 1500-step limit) x 64 steps per rollout, shared tanh trunk "
                       "[128, 64], 10 epochs x 16 minibatches of 65 536 samples, random initial weights"}
    try:
        env = BatchedRocket6DOF(n, default_env_config(), remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, seed=1, device=dev)
        algo = PPO(env, PPOConfig(n_steps=steps, batch_size=1 << 16, net_arch=(128, 64), shared_trunk=True, log_std_min=-2.5))
        for _ in range(warm):
            algo.collect_rollouts()
            algo.train()
        torch.cuda.synchronize()
        roll = upd = 0.0
        for _ in range(iters):
            t0 = time.perf_counter()
            algo.collect_rollouts()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            algo.train()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            roll += t1 - t0
            upd += t2 - t1
        roll, upd = roll / iters, upd / iters
        out.update({"env_steps_per_s": steps * n / (roll + upd), "rollout_and_gae_ms": roll * 1e3, "update_ms": upd * 1e3,
                    "minibatch_step_us": upd / 160 * 1e6, "iterations_timed": iters})
        del algo, env
        torch.cuda.empty_cache()
    except Exception as e:   # noqa: BLE001 -- a side figure must not cost the line
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    return out


def ppo_configs4_side_measurement(torch, dev, iters: int = 2, warm: int = 2) -> dict:
    """The ONE-GPU share of BASELINE configs[4] ("1 048 576 envs x 8 GPUs + SB3 PPO MLP(64,64), RCCL all-gather of returns"): 2^20
    wrapped envs (the reference's make_env() stack, main_6DOF.py:44-53), SB3's default topology -- two separate tanh MLPs (64, 64)
    (test_6DOF_sb_integration.py:31) -- 64 steps per rollout, 10 epochs of 1024 minibatches of 65 536 samples, and the design's one
    collective per rollout, the gather of the episode summary (a local copy with one rank; RCCL with several).  What an 8-GPU job
    adds per rank: the gradient all-reduce inside every minibatch step, captured into the same graph
    (tests/test_gpu_ppo.py::test_multi_rank_update_runs_from_a_graph_with_the_all_reduce_inside).  A side figure of the line."""
    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
    from rl_rocket_6dof_amd import dist as r6dist
    from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

    n, steps = 1 << 20, 64
    out = {"workload": "PPO, one GPU's share of BASELINE configs[4]: 2^20 wrapped envs (remove-mass obs, reward clip, 1500-step limit) x 64 steps per "
                       "rollout, two separate tanh MLPs (64, 64), 10 epochs x 1024 minibatches of 65 536 samples, one episode-summary gather per "
                       "rollout, random initial weights"}

    def run(bs: int) -> dict:
        env = BatchedRocket6DOF(n, default_env_config(), remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, seed=1, device=dev)
        algo = PPO(env, PPOConfig(n_steps=steps, batch_size=bs, net_arch=(64, 64), shared_trunk=False, log_std_min=-2.5))
        gathered = []
        for _ in range(warm):
            algo.collect_rollouts()
            gathered.append(r6dist.gather_episode_stats_device(env.episode_stats_tensor()))
            algo.train()
        torch.cuda.synchronize()
        roll = upd = 0.0
        for _ in range(iters):
            t0 = time.perf_counter()
            algo.collect_rollouts()
            gathered.append(r6dist.gather_episode_stats_device(env.episode_stats_tensor()))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            algo.train()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            roll += t1 - t0
            upd += t2 - t1
        roll, upd = roll / iters, upd / iters
        mb = 10 * (steps * n // bs)
        rec = {"env_steps_per_s": steps * n / (roll + upd), "rollout_gae_and_gather_ms": roll * 1e3, "update_ms": upd * 1e3,
               "minibatch_steps_per_iteration": mb, "minibatch_step_us": upd / mb * 1e6, "iterations_timed": iters,
               "rollout_us_per_env_step_of_the_batch": roll / steps * 1e6, "fused_update": algo.fused_update is not None,
               "update_from_graph": algo._graph is not None, "episodes_in_last_gather": float(gathered[-1][:, 0].sum())}
        del algo, env, gathered
        torch.cuda.empty_cache()
        return rec

    try:
        out.update(run(1 << 16))
        # the same job with SB3's own ratio of minibatches to rollout (PPO defaults: 2048 samples / 64 = 32 minibatches per epoch; here 64 of
        # 2^20 samples): the update's fixed cost per minibatch step (weight images, partial sets, reduce, clip + Adam) is paid 16x less often
        big = run(1 << 20)
        big["workload"] = "as above with 64 minibatches of 1 048 576 samples per epoch"
        out["minibatch_2^20"] = big
    except Exception as e:   # noqa: BLE001 -- a side figure must not cost the line
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    return out


def rollout_gather_loop(env_g, pool, args, barrier, stream, torch, r6dist, dev) -> dict:
    """The N > 1 record's second loop, itemised (rank-reduced with max over ranks):
      first_gather_us   the very first all-gather of the process (communicator creation, buffer registration) timed alone,
                        outside every other number;
      then one UNTIMED pass with two gathers (warm), then, on the same env and the same number of steps,
      wall without / with a gather every `gather_every` steps  ->  us_per_step_tracked_env, us_per_gather (the difference per
      gather) and `value` (the loop with gathers, whole job)."""
    ge = args.gather_every
    gsteps = max(2 * ge, min(args.steps, 10 * ge) // ge * ge)
    if args.k > 1:
        gsteps = max(gsteps // args.k * args.k, args.k)
    gathers = []     # per-rank tables stay on the device: the all-gather is stream-ordered, the loop never drains the GPU
    do_gather = lambda: gathers.append(r6dist.gather_episode_stats_device(env_g.episode_stats_tensor()))
    barrier()
    t0 = time.perf_counter()
    do_gather()
    torch.cuda.synchronize(dev)
    first_us = (time.perf_counter() - t0) * 1e6
    timed_loop(env_g, pool, 2 * ge if args.k <= 1 else max(2 * ge // args.k * args.k, args.k), args.k, barrier, stream, torch,
               every=(ge, do_gather))                                                        # warm: >= 2 gathers, untimed
    wall_plain, _ = timed_loop(env_g, pool, gsteps, args.k, barrier, stream, torch)
    n_before = len(gathers)
    wall_g, _ = timed_loop(env_g, pool, gsteps, args.k, barrier, stream, torch, every=(ge, do_gather))
    n_g = len(gathers) - n_before
    red_p = r6dist.reduce_bench_times(wall_plain, first_us, device=dev)      # (the kernel_ms slot carries first_us)
    red_g = r6dist.reduce_bench_times(wall_g, 0.0, device=dev)
    wp, wg = red_p["wall_max"], red_g["wall_max"]
    return {"gather_every": ge, "steps": gsteps, "gathers": n_g, "wall_s": wg, "unit": "env-steps/s",
            "us_per_step_tracked_env": wp / gsteps * 1e6, "us_per_step_with_gathers": wg / gsteps * 1e6,
            "us_per_gather": (wg - wp) / max(n_g, 1) * 1e6, "first_gather_us": red_p["kernel_ms_max"],
            "warm": "one untimed pass with 2 gathers before both timed loops; same env, same steps",
            "episodes_seen": float(sum(float(t[:, 0].sum()) for t in gathers[n_before:]))}


# what every rank of a self-launched job runs (tests point the ATTRIBUTE, or launch_ranks(script=...), at a GPU-free stand-in; there
# is no environment seam: a variable left set would have made `python bench.py --gpus N` print a stand-in's line -- round-4 advice)
RANK_SCRIPT = os.path.abspath(__file__)


class _stdout_to_stderr:
    """File descriptor 1 points at stderr for the duration (native code that writes to stdout, e.g. gloo's connection banner)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks: int, argv, script: str = None) -> int:
    """`python bench.py --gpus N` from a plain shell (no WORLD_SIZE in the environment): this process becomes the LAUNCHER.  It
    never imports torch.cuda nor touches a GPU; it starts

        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same args>

    as a CHILD process (no os.exec*: a GPU box refuses an exec from a process that has initialised the GPU, and the launcher
    has to outlive the job to report on it), relays the child's stdout -- rank 0's ONE JSON line -- and returns the child's
    return code.  Loud when it cannot: a child that fails, or that ends without a line whose n_gpus == N, makes the launcher
    exit non-zero with the reason on stderr.  The reference has no launcher to mirror (main_6DOF.py:155-156 is one process,
    n_envs = 1); the job it replaces is the loop of main_6DOF.py:105-139."""
    script = script or RANK_SCRIPT
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    env["R6_BENCH_LAUNCHER_PID"] = str(os.getpid())
    print(f"[bench] launcher: {n_ranks} ranks, one per GPU: {' '.join(cmd)}", file=sys.stderr, flush=True)
    def _die_with_parent():      # Linux: the job gets SIGTERM when this launcher dies (a driver's timeout kills the launcher only:
        try:                     # the ranks live in their own session); torch.distributed.run then ends its workers
            import ctypes
            import signal

            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGTERM))      # PR_SET_PDEATHSIG
        except Exception:
            pass

    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True,   # own process group
                            preexec_fn=_die_with_parent)
    seen = seen_ln = None
    try:
        for ln in proc.stdout:       # rank 0's line is HELD until the job's return code is known (a rank that crashes in the teardown
            rec = None               # after the line was printed must not leave a well-formed record on stdout -- round-4 advice);
            if ln.lstrip().startswith("{"):    # anything else a rank or a library prints there ("[Gloo] Rank 0 is connected ...")
                try:                           # goes to stderr: the launcher's stdout is the ONE line
                    rec = json.loads(ln)
                except ValueError:
                    rec = None
            if isinstance(rec, dict) and "n_gpus" in rec:
                if seen_ln is not None:
                    sys.stderr.write("[bench] launcher: superseded line: " + seen_ln)
                seen, seen_ln = rec, ln
            else:
                sys.stderr.write(ln)
                sys.stderr.flush()
        rc = proc.wait()
    except BaseException:
        if proc.poll() is None:      # interrupted: end the exact process group this launcher started (torchrun and its ranks)
            import signal

            try:
                os.killpg(proc.pid, signal.SIGTERM)
                proc.wait(timeout=20)
            except (OSError, subprocess.TimeoutExpired):
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
        raise
    if rc != 0:
        if seen_ln is not None:
            sys.stderr.write("[bench] launcher: discarded (the job failed after printing it): " + seen_ln)
        print(f"[bench] launcher: the {n_ranks}-rank job FAILED (torch.distributed.run returned {rc}); no scaling record", file=sys.stderr)
        return rc if 0 < rc < 256 else 1
    if seen is None or seen.get("n_gpus") != n_ranks:
        got = "no JSON line" if seen is None else f"a line with n_gpus = {seen.get('n_gpus')}"
        if seen_ln is not None:
            sys.stderr.write("[bench] launcher: discarded (wrong world): " + seen_ln)
        print(f"[bench] launcher: asked for {n_ranks} ranks but the job printed {got}", file=sys.stderr)
        return 3
    sys.stdout.write(seen_ln)
    sys.stdout.flush()
    return 0


def check_world(args) -> "int | None":
    """How many ranks the environment says there are against --gpus.  None: not launched yet (no WORLD_SIZE) and --gpus > 1 --
    this process must become the launcher.  A WORLD_SIZE that disagrees with --gpus ends the run, whatever the two numbers
    (round 3 raised only for WORLD_SIZE > 1: `--gpus 8` from a plain shell stepped one GPU and printed n_gpus 1)."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is None:
        return None if args.gpus > 1 else 1
    try:
        world = int(ws)
    except ValueError:
        raise SystemExit(f"bench.py: WORLD_SIZE={ws!r} is not a number")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` from a plain "
                         f"shell (it launches its own ranks) or under torch.distributed.run with --nproc-per-node {args.gpus}")
    return world


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.k > 1 and args.steps % args.k:
        raise SystemExit("--steps must be a multiple of --k")
    if check_world(args) is None:
        # BEFORE anything imports torch.cuda: the launcher process never owns a GPU context
        rc = launch_ranks(args.gpus, list(sys.argv[1:] if argv is None else argv))
        if rc != 0:
            raise SystemExit(rc)
        return

    import torch
    import torch.distributed as dist

    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
    from rl_rocket_6dof_amd import dist as r6dist

    # R6_DIST_BACKEND=gloo + R6_FORCE_DEVICE=0 rehearse the N>1 code path with several ranks on ONE GPU
    # (RCCL refuses two ranks per device); the driver's real runs use the defaults (nccl, LOCAL_RANK).
    # From here to the JSON line, file descriptor 1 points at stderr: native code announces itself on stdout (gloo's "[Gloo] Rank r is
    # connected ...", RCCL's five-line version banner at its first collective), and this job's stdout carries the ONE line and nothing else
    quiet = _stdout_to_stderr().__enter__()
    rank, world, local = r6dist.init_from_env(backend=os.environ.get("R6_DIST_BACKEND"))
    # R6_FORCE_PROCESS_GROUP=1: a process group of ONE rank (backend nccl = RCCL by default) and the N > 1 code path on it --
    # every collective of that path (device barrier, all-gathers of times and episode summaries) then really goes through
    # RCCL on the one GPU a builder has; the numbers are those of a 1-GPU run
    force_pg = world == 1 and os.environ.get("R6_FORCE_PROCESS_GROUP") == "1"
    if force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("R6_FORCE_DEVICE", local)))
        dist.init_process_group(backend=os.environ.get("R6_DIST_BACKEND") or "nccl", rank=0, world_size=1)
    if world != args.gpus:     # check_world() above saw the same WORLD_SIZE: cannot happen, and must not pass silently if it does
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product has no CPU path)")
    local = int(os.environ.get("R6_FORCE_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    backend = dist.get_backend() if (world > 1 or force_pg) else None
    on_gpu_pg = backend == "nccl"

    n = args.envs_per_gpu
    cfg = default_env_config()
    wrapped = args.variant == "wrapped"
    multi = world > 1 or force_pg
    env = BatchedRocket6DOF(
        n, cfg, device=dev, integrator=args.integrator, timestep=args.dt, env_offset=rank * n, auto_reset=True,
        record_flags=False, record_terms=False, track_episodes=wrapped,     # the headline workload is the same at every N
        remove_mass=wrapped, clip_reward=(-1, 100) if wrapped else None,
        max_episode_steps=int(150 / (args.dt or cfg["timestep"])) if wrapped else 0, seed=42)
    pool = make_pool(torch, n, args.k, args.pool, dev, 1234 + rank)
    env.reset()
    if args.k <= 1:
        for j in range(args.burnin):
            env.step_async(pool[j % args.pool])
    else:
        for j in range(max(1, args.burnin // args.k)):
            env.step_k(pool[j % args.pool], last_only=True)
        env._bench_out = env.step_k(pool[0])

    # The barriers around the HEADLINE loop go through a gloo group: a live RCCL communicator on the device makes the very
    # same step kernel 2 us slower per launch (37.5 -> 39.6 us, measured with one rank in one process, R6_FORCE_PROCESS_GROUP=1),
    # and torch creates the communicator at the first RCCL collective -- so RCCL first runs after the headline's timed region
    # (the reductions of the times, the rollout-gather loop and its barriers), and the raw step is timed alike at every N.
    host_group, rccl_live = None, [False]
    if multi and on_gpu_pg:
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node: do not depend on the hostname resolving
        try:
            host_group = dist.new_group(backend="gloo")
        except Exception as e:
            print(f"[bench] rank {rank}: gloo group unavailable ({e})", file=sys.stderr)
            host_group = None
        # one decision for the whole job: a rank that waits on the gloo barrier while another one waits on an RCCL barrier
        # hangs inside the timed region.  The ranks exchange an ok flag through the process group's key-value store (TCP, not
        # a collective of either backend) and take gloo only if every rank has it.
        if not r6dist.all_ranks_agree(host_group is not None, "bench_gloo_ok"):
            if rank == 0:
                print("[bench] no gloo group on some rank: barriers through RCCL on every rank", file=sys.stderr)
            host_group = None
            rccl_live[0] = True

    def barrier():
        torch.cuda.synchronize(dev)
        if multi:
            if on_gpu_pg and rccl_live[0]:
                dist.barrier(device_ids=[local])     # RCCL: name the device, or the barrier guesses it from the rank
            elif on_gpu_pg:
                dist.barrier(group=host_group)
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    stream = torch.cuda.current_stream(dev)
    if args.preroll_seconds > 0:
        # the ranks start their pre-roll together: it runs by wall time, and a rank that got here 0.3 s early (its env was built
        # sooner) would sit idle in the next barrier for as long -- 20 ms of idleness are enough for the chip to clock down, and that
        # rank's timed region (the job's time is the slowest rank's) would then run at 39-45 us per step
        barrier()
        preroll(env, pool, args.preroll_seconds, args.k, lambda: torch.cuda.synchronize(dev), args.preroll_batch)     # untimed, by time
    timed_loop(env, pool, max(args.warmup, args.k), args.k, barrier, stream, torch)        # W untimed steps
    wall, kernel_ms = timed_loop(env, pool, args.steps, args.k, barrier, stream, torch)    # EXACTLY K timed steps

    extras = {}
    if args.steady_launches > 0:
        # the distribution the timed region above was drawn from, same env, same process, right behind it (every rank takes it:
        # the ranks stay in step; the line carries rank 0's)
        st = steady_record(env, pool, args.steady_launches, args.steady_chunk, args.k, stream, torch, lambda: torch.cuda.synchronize(dev))
        st["timed_region_kernel_us"] = kernel_ms * 1e3
        st["timed_region_over_median"] = kernel_ms * 1e3 / st["median"]
        st["note"] = ("the contract's timed region is ONE sample of --steps launches; at --steps 20 it is 0.8 ms long, shorter than a slow "
                      "episode of the chip, so its value moves within [min, max] of these chunk means from run to run; wall-clock "
                      "ms_per_step adds the start-up of the first launch after the barrier and the host's wake-up (~1.7 us per step at 20 steps)")
        extras["steady"] = st
    if multi:
        # the job's time = the slowest rank's (host floats: over the gloo group when there is one, so that the record of the timed
        # region does not depend on the device collectives coming up afterwards)
        red = r6dist.reduce_bench_times(wall, kernel_ms, device=dev, group=host_group)
        wall, kernel_ms = red["wall_max"], red["kernel_ms_max"]
        extras["rccl"] = {"backend": backend, "rccl_ranks": dist.get_world_size(), "devices": red["devices"],
                          "kernel_us_min": red["kernel_ms_min"] * 1e3, "kernel_us_max": red["kernel_ms_max"] * 1e3,
                          "wall_s_min": red["wall_min"], "wall_s_max": red["wall_max"],
                          "verified_with_more_than_one_rank": dist.get_world_size() > 1}
        # the one collective of the design (episode summaries) inside a second, short timed loop:
        # BASELINE configs[4]'s per-rollout exchange.  Its own env, with the episode accounting the summaries come from
        # (ep_step / ep_return rows, +16 B per env-step and the statistics): the headline loop above stays the raw step at
        # every N, so that the driver's scaling ratios compare like with like.  From here on the barriers go through the
        # device backend too (RCCL's communicator comes up at its first collective).  An error of the device collectives that
        # every rank sees (RCCL not coming up on this node) is recorded in the line instead of costing the job its record.
        try:
            rccl_live[0] = True
            env_g = env if wrapped else BatchedRocket6DOF(
                n, cfg, device=dev, integrator=args.integrator, timestep=args.dt, env_offset=rank * n, auto_reset=True,
                record_flags=False, record_terms=False, track_episodes=True, seed=42)
            if env_g is not env:
                env_g.reset()
                if args.k <= 1:
                    for j in range(args.burnin):
                        env_g.step_async(pool[j % args.pool])
                else:
                    for j in range(max(1, args.burnin // args.k)):
                        env_g.step_k(pool[j % args.pool], last_only=True)
                    env_g._bench_out = env_g.step_k(pool[0])
            gather = rollout_gather_loop(env_g, pool, args, barrier, stream, torch, r6dist, dev)
            gather["value"] = n * world * gather["steps"] / gather.pop("wall_s")
            extras["rccl"]["with_rollout_gather"] = gather
        except Exception as e:      # noqa: BLE001 -- whatever the backend raises
            print(f"[bench] rank {rank}: rollout-gather loop failed: {type(e).__name__}: {e}", file=sys.stderr)
            extras["rccl"]["with_rollout_gather"] = {"error": f"{type(e).__name__}: {e}"[:600]}
            rccl_live[0] = host_group is None      # the closing barrier over gloo when there is a gloo group

    if rank == 0:
        dt = args.dt or cfg["timestep"]
        traffic, source = traffic_from_profile(args.variant, args.integrator, n, args.k)
        valu = valu_from_profile(args.variant, args.integrator, n, args.k, kernel_ms)
        if valu is not None:
            extras["valu"] = valu
        if not multi and not args.no_extras:
            del env, pool
            torch.cuda.empty_cache()
            extras["out_of_cache"] = side_measurement(
                torch, dev, cfg, n=1 << 23, integrator="rk4", dt=None, k=1, pool_count=8, burnin=600, steps=200, warmup=40,
                label="2^23 envs, RK4 dt 0.1, raw step, 8 cold action buffers: same kernel, 1.6 GB working set (HBM, not the Infinity Cache)")
            extras["configs1"] = {
                "one_launch_per_step": side_measurement(
                    torch, dev, cfg, n=65536, integrator="euler", dt=0.04, k=1, pool_count=2, burnin=2000, steps=4000,
                    warmup=400, label="BASELINE configs[1]: 65 536 envs, Euler dt 0.04, r6_step per step"),
                "k16": side_measurement(
                    torch, dev, cfg, n=65536, integrator="euler", dt=0.04, k=16, pool_count=2, burnin=2000, steps=8000,
                    warmup=800, label="BASELINE configs[1]: 65 536 envs, Euler dt 0.04, r6_step_k with 16 steps per launch, "
                                      "all per-step outputs stored")}
            extras["rk45"] = side_measurement(
                torch, dev, cfg, n=1 << 20, integrator="rk45", dt=None, k=1, pool_count=2, burnin=600, steps=200, warmup=40,
                label="2^20 envs, in-kernel adaptive RK45 (scipy's step control, the reference's own integrator: simulator.py:83-88), dt 0.1, raw step: "
                      "bound by arithmetic (8-20 RHS evaluations per env step, the wave follows its slowest lane), not by HBM")
            extras["two_groups"] = two_groups_side_measurement(torch, dev, cfg)
            extras["configs0"] = single_env_side_measurement(torch, dev)
            extras["ppo_training"] = ppo_side_measurement(torch, dev)
            extras["ppo_configs4"] = ppo_configs4_side_measurement(torch, dev)
        if not multi and not args.no_cpu_baseline and args.cpu_seconds > 0:
            extras["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        line = build_line(args, world, n, dt, wall, kernel_ms, traffic=traffic, traffic_source=source, extras=extras)
        quiet.__exit__()
        print(json.dumps(line), flush=True)
        quiet.__enter__()         # the teardown below may talk as well

    if multi:
        try:
            barrier()
            dist.destroy_process_group()
        except Exception as e:      # noqa: BLE001 -- the line is out; a teardown error must not turn the job's exit code
            print(f"[bench] rank {rank}: teardown: {type(e).__name__}: {e}", file=sys.stderr)
    quiet.__exit__()


if __name__ == "__main__":
    main()
