"""`python bench.py --gpus 2` from a plain shell on the GPU box: the launcher branch with REAL ranks (round-3 review, item 1).  Two ranks
share the one GPU a builder's box has (R6_FORCE_DEVICE=0) and talk over gloo (RCCL refuses two ranks on one device); everything
else -- the child torch.distributed.run, the env shards with their global offsets, the timed loops between barriers, the reduction
of the per-rank times, the rollout-gather loop -- is what an 8-GPU run executes.  The loop served: main_6DOF.py:105-139."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_gpus_2_launches_two_ranks_on_this_gpu():
    env = dict(os.environ, R6_DIST_BACKEND="gloo", R6_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs-per-gpu", "65536", "--steps", "64", "--warmup", "16",
                        "--burnin", "200", "--gather-every", "16"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip() == lines[0], "the launcher's stdout is the ONE JSON line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["envs_total"] == 2 * 65536 and line["scaling"] == "weak"
    rc = line["rccl"]
    assert rc["rccl_ranks"] == 2 and rc["backend"] == "gloo" and rc["devices"] == [0, 0] and rc["verified_with_more_than_one_rank"] is True
    assert rc["with_rollout_gather"]["gathers"] >= 2 and rc["with_rollout_gather"]["episodes_seen"] > 0
    assert line["value"] > 1e8 and line["roofline"]["kernel_us"] > 0
    assert "cpu_baseline" not in line and "out_of_cache" not in line          # N = 1 extras stay out of the N > 1 record


def test_ranks_started_by_torch_distributed_run_print_one_line():
    """The DRIVER's way of starting an N > 1 job: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, no launcher
    of ours in between -- the ranks' own stdout is what gets parsed.  Native code announces itself there (gloo's "[Gloo] Rank r is
    connected ...", RCCL's version banner at its first collective): bench.py points file descriptor 1 at stderr from the first
    process-group call to the line itself, so the job's stdout is the ONE JSON line.  Two ranks share the one GPU over gloo."""
    env = dict(os.environ, R6_DIST_BACKEND="gloo", R6_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import socket
    slack = []
    for attempt in range(3):       # the timing bound below is about a host-side latency: one quiet attempt of three is enough
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs-per-gpu", "65536", "--steps", "20",
                            "--warmup", "5", "--burnin", "200", "--preroll-seconds", "0.2", "--steady-launches", "0"],
                           env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        out = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(out) == 1 and out[0].startswith("{"), r.stdout[-1500:]
        line = json.loads(out[0])
        assert line["n_gpus"] == 2 and line["steps"] == 20 and line["rccl"]["rccl_ranks"] == 2
        # a rank's clock stops at its own synchronize, in front of the closing barrier: the job's wall time is the slowest rank's, within
        # the first launch's start-up and the host's wake-up of its kernel time -- not a host-side barrier (>= 100 us) on top of it
        slack.append(line["rccl"]["wall_s_max"] - 20 * line["rccl"]["kernel_us_max"] * 1e-6)
        if slack[-1] < 90e-6:
            break
    assert min(slack) < 90e-6, slack


def test_bench_single_gpu_prints_one_contract_line():
    """`python bench.py` the way the driver runs it at N = 1 (reduced sizes): ONE JSON line on stdout with the contract's keys, the
    roofline and cpu_baseline objects, and the side figures of this round (rk45, valu at the measured issue rates)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "20", "--burnin", "300", "--cpu-seconds", "1.5"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 100 and line["dtype"] == "f32" and line["vs_baseline"] is None
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.4 < rf["frac"] < 0.9 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] and rf["traffic"] > 1.9e8 and "static file" in rf["traffic_source"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 100 and "sample" in cb
    assert 60 < line["rk45"]["us_per_env_step"] < 300 and line["rk45"]["integrator"] == "rk45"
    assert "issue_time_us_per_simd" in line["valu"] and "busy_frac" not in line["valu"]
    assert line["configs1"]["k16"]["roofline"]["bound"] == "dependent_chain_of_one_tile_per_simd"
    c0 = line["configs0"]           # one env driven from the host: faster than the reference's Python (730 steps/s on one Xeon core)
    assert c0["rocket6dof_rk4_steps_per_s"] > 2000 and c0["rocket6dof_rk45_steps_per_s"] > 2000 and c0["simulator6dof_rk4_steps_per_s"] > 2000


def test_driver_shape_20_steps_sits_in_its_own_steady_distribution():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- the driver's exact command shape (round-4 review, Next 1; side figures and
    CPU baseline switched off here, they run after the timed region and do not touch it).  What round 5 measured about it
    (DESIGN.md section 6): the 20 launches last 0.8 ms; the chip has slow episodes of 1.5-3 ms at +5 ... +15 % that arrive every
    8-25 ms, do NOT replay with the data (the same deterministic sequence run twice: 41 and 34 slow 20-launch chunks, 1 in common) and
    hit 15-25 % of all 20-launch windows.  So one 20-launch sample cannot be held to 4 % of the steady median every time -- what can
    be asserted, and is:
      * every run: the timed region lies inside the [min, max] of the 100 chunk means of the `steady` record taken right behind
        it in the same process (it is a sample of that distribution, nothing else is going on), and its wall-clock figure exceeds
        its HIP-event figure by no more than the 20-step shape's fixed costs (first launch after the barrier + host wake-up);
      * of at most four runs, one has its 20-step kernel mean between the fast end of its own 2000-launch steady record (0.97 x the
        p10 of the chunk means: the median itself is pulled up by the slow chunks, and a 20-launch sample in a quiet stretch reads
        0.95-0.98 x the median -- three of four runs of one box did) and 1.06 x the median (4 % + the shape's own +0.8 us, the
        first launch behind the barrier, in the HIP-event figure); four misses in a row have a probability below 0.5 % at the
        measured share of clock dips."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    attempts, ok = [], False
    for _ in range(4):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras",
                            "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        st = line["steady"]
        assert line["steps"] == 20 and line["warmup"] == 5 and st["chunks"] == 100 and st["chunk_launches"] == 20
        assert "1 s of untimed stepping" in line["config"]["workload"]
        k_us, wall_us = line["roofline"]["kernel_us"], line["ms_per_step"] * 1e3
        attempts.append({"kernel_us": k_us, "wall_us": wall_us, "steady": {k: st[k] for k in ("median", "p10", "p90", "min", "max", "mean")},
                         "slow_share": st["share_of_chunks_above_1.06_x_median"]})
        assert st["min"] * 0.97 <= k_us <= st["max"] * 1.08, attempts[-1]      # (a dip deeper than any of the 100 steady chunks happens to hold: seen up to 1.02 x max)
        assert 0.0 <= wall_us - k_us < 3.0, attempts[-1]            # measured: +0.9 ... +1.0 us per step at 20 steps
        assert 0.45 < line["roofline"]["frac"] < 0.8          # a 20-launch sample inside a deep clock dip has read 45-47 us (0.53-0.55)
        if st["p10"] * 0.97 <= k_us <= st["median"] * 1.06:
            ok = True
            break
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(attempts, open(os.path.join(ROOT, "gpurun_out", "driver_shape_attempts.json"), "w"), indent=1)
    except OSError:
        pass
    print(f"[driver shape] {len(attempts)} run(s): " + "; ".join(f"20-step {a['kernel_us']:.2f} us vs steady median {a['steady']['median']:.2f} "
                                                                 f"(p10 {a['steady']['p10']:.2f}, p90 {a['steady']['p90']:.2f})" for a in attempts))
    assert ok, attempts
