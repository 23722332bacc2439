"""bench.py's argument handling and line building without a GPU: the N > 1 branch (rank reduction over gloo with world
size 2, the `rccl` object, whole-job value) must not meet its first typo on the 8-GPU box."""
import json
import os
import socket
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_line_contract_single_gpu():
    args = bench.parse_args([])
    assert (args.gpus, args.steps, args.warmup, args.k) == (1, 2000, 200, 1)
    n = args.envs_per_gpu
    line = bench.build_line(args, 1, n, 0.1, wall=2000 * 38e-6, kernel_ms=0.0375, traffic=2.04e8,
                            traffic_source="profiles/x.json (static file)", extras={"cpu_baseline": {"value": 2300.0}})
    json.loads(json.dumps(line))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["vs_baseline"] is None and line["dtype"] == "f32"
    assert abs(line["value"] - n / 38e-6) < 1e-3 * line["value"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["achieved"] - 189 * n / 37.5e-6 / 1e9) < 1e-6 * r["achieved"] and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["traffic_source"].startswith("profiles/") and abs(r["kernel_us"] - 37.5) < 1e-9
    assert "model" not in line["config"] and "workload" in line["config"]
    # record hygiene (round 3): the library's build stamp and this file's digest, each named for what it is
    assert "commit" not in line["config"] and "lib_commit" in line["config"]
    import hashlib
    assert line["config"]["bench_py_sha16"] == hashlib.sha256(open(os.path.join(ROOT, "bench.py"), "rb").read()).hexdigest()[:16]


def test_bytes_per_env_step_formula():
    assert bench.bytes_per_env_step("raw") == 189 and bench.bytes_per_env_step("wrapped") == 201
    # K steps per launch: action 12 + obs 56 + reward 4 + done 1 per step, state 2 x 56 + epconst 2 x 8 once per K steps
    assert bench.bytes_per_env_step("raw", 16) == pytest.approx(73 + 128 / 16)
    assert bench.bytes_per_env_step("raw", 1 << 30) == pytest.approx(73, rel=1e-6)


def test_k_must_divide_steps():
    with pytest.raises(SystemExit):
        bench.main(["--k", "16", "--steps", "100"])


STUB = os.path.join(ROOT, "tests", "_bench_rank_stub.py")


def _no_dist_env(monkeypatch):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)


def test_gpus_n_from_a_plain_shell_launches_n_ranks(monkeypatch, capfd):
    """Round-3 review, item 1: `python bench.py --gpus 2` with no WORLD_SIZE must become a 2-rank job by itself (a child
    torch.distributed.run), relay rank 0's line and exit 0 -- not step one GPU and print n_gpus 1.  The ranks here are
    GPU-free stand-ins over gloo (tests/_bench_rank_stub.py); the launcher, the argument relay and the checks are bench.py's."""
    _no_dist_env(monkeypatch)
    monkeypatch.setattr(bench, "RANK_SCRIPT", STUB)
    bench.main(["--gpus", "2", "--steps", "2000", "--warmup", "100"])
    out, err = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl"]["rccl_ranks"] == 2 and line["rccl"]["backend"] == "gloo"
    assert line["steps"] == 2000 and line["warmup"] == 100                      # the same arguments reached the ranks
    assert line["value"] == pytest.approx(2 * (1 << 20) * 2000 / 0.084)        # whole job over the slowest rank
    assert line["rccl"]["with_rollout_gather"]["episodes_seen"] == 3.0
    assert "torch.distributed.run" in err and "--nproc-per-node=2" in err


def test_launcher_is_loud_when_a_rank_fails_or_the_world_is_wrong(monkeypatch, capfd):
    _no_dist_env(monkeypatch)
    monkeypatch.setattr(bench, "RANK_SCRIPT", STUB)
    monkeypatch.setenv("R6_STUB_FAIL_RANK", "1")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert e.value.code not in (0, None)
    assert "FAILED" in capfd.readouterr().err
    monkeypatch.delenv("R6_STUB_FAIL_RANK")
    # a rank that fails AFTER rank 0 has printed its line (teardown crash): the held line must not reach stdout (round-4 advice)
    monkeypatch.setenv("R6_STUB_FAIL_AFTER_LINE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    out, err = capfd.readouterr()
    assert e.value.code not in (0, None) and "discarded" in err and not [ln for ln in out.splitlines() if ln.startswith("{")], out
    monkeypatch.delenv("R6_STUB_FAIL_AFTER_LINE")
    monkeypatch.setenv("R6_STUB_WRONG_WORLD", "1")       # a job that ends well but reports the wrong number of ranks
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    out, err = capfd.readouterr()
    assert e.value.code == 3 and "n_gpus = 1" in err and not [ln for ln in out.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("world,gpus", [("1", "8"), ("2", "1"), ("4", "8"), ("x", "1")])
def test_world_size_that_disagrees_with_gpus_ends_the_run(monkeypatch, world, gpus):
    """... in every case, world == 1 included (round 3: `WORLD_SIZE=1 bench.py --gpus 8` ran one rank and said n_gpus 1)."""
    monkeypatch.setenv("WORLD_SIZE", world)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", gpus])
    assert e.value.code not in (0, None)


def test_launcher_process_never_loads_the_gpu_runtime():
    """The launcher must not own a GPU context (the ranks are its children): a fresh interpreter that takes the launcher branch
    has not imported torch when launch_ranks() is reached."""
    import subprocess

    code = ("import sys, os; sys.path.insert(0, %r); os.environ.pop('WORLD_SIZE', None); import bench\n"
            "def fake(n, argv, script=None):\n"
            "    assert 'torch' not in sys.modules, 'launcher imported torch'\n"
            "    print('LAUNCH', n, argv); return 0\n"
            "bench.launch_ranks = fake; bench.main(['--gpus', '8', '--steps', '5'])" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "LAUNCH 8 ['--gpus', '8', '--steps', '5']" in r.stdout


def test_ranks_do_not_outlive_a_killed_launcher(tmp_path):
    """A driver that times the bench out kills the LAUNCHER; the ranks live in a session of their own (so that an interrupted launcher
    can end exactly its own process group) and must not keep their GPUs: the child torch.distributed.run is started with
    PR_SET_PDEATHSIG and ends its workers when the launcher dies."""
    import signal
    import subprocess
    import time

    env = dict(os.environ, R6_STUB_PIDDIR=str(tmp_path), R6_STUB_SLEEP="120")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # a launcher in a process of its own: a small driver that calls bench.launch_ranks(script=STUB) (bench.py has no environment seam)
    code = "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_ranks(2, ['--gpus', '2'], script=%r))" % (ROOT, STUB)
    launcher = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        t0 = time.time()
        while len(list(tmp_path.glob("rank*.pid"))) < 2:
            assert time.time() - t0 < 120 and launcher.poll() is None, "the stand-in ranks did not start"
            time.sleep(0.2)
        pids = [int(p.read_text()) for p in tmp_path.glob("rank*.pid")]
        launcher.send_signal(signal.SIGKILL)
        launcher.wait(timeout=30)

        def alive(pid):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                return False
            try:     # a zombie whose parent has not reaped it yet counts as gone
                return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
            except OSError:
                return False

        t0 = time.time()
        while any(alive(p) for p in pids):
            assert time.time() - t0 < 60, f"ranks {pids} outlived their launcher"
            time.sleep(0.5)
    finally:
        if launcher.poll() is None:
            launcher.kill()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench as b
    from rl_rocket_6dof_amd import dist as r6dist

    r, w, _ = r6dist.init_from_env(backend="gloo")
    # what main() does after its timed loops, with the GPU part replaced by fixed per-rank numbers
    wall, kernel_ms = (0.080, 0.0380) if rank == 0 else (0.084, 0.0402)
    red = r6dist.reduce_bench_times(wall, kernel_ms)
    # the same reduction through an explicit host group (what bench.py does when the default backend is RCCL: the record of the
    # timed region must not wait for the device collectives) gives the same numbers
    host_group = dist.new_group(backend="gloo")
    assert r6dist.reduce_bench_times(wall, kernel_ms, group=host_group) == red
    # the gloo-or-RCCL decision of bench.py is a GROUP decision: one rank without a gloo group switches every rank
    assert r6dist.all_ranks_agree(True, "t1") is True
    assert r6dist.all_ranks_agree(rank != 1, "t2") is False
    # the same key again with other answers: every call is its own generation, nothing of the first call is read back
    assert r6dist.all_ranks_agree(True, "t2") is True
    assert r6dist.all_ranks_agree(rank != 0, "t1") is False
    tot, _ = r6dist.gather_episode_stats({k: float(rank + 1) for k in r6dist.STAT_NAMES})
    args = b.parse_args(["--gpus", "2", "--steps", "2000", "--warmup", "100"])
    extras = {"rccl": {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "devices": red["devices"],
                       "kernel_us_min": red["kernel_ms_min"] * 1e3, "kernel_us_max": red["kernel_ms_max"] * 1e3,
                       "with_rollout_gather": {"gather_every": args.gather_every, "episodes_seen": tot["episodes"]}}}
    line = b.build_line(args, w, args.envs_per_gpu, 0.1, red["wall_max"], red["kernel_ms_max"], extras=extras)
    q.put((rank, json.dumps(line)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_line_over_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    l0, l1 = json.loads(res[0]), json.loads(res[1])
    assert l0["n_gpus"] == 2 and l0["config"]["envs_total"] == 2 * (1 << 20)
    # whole-job value over the SLOWEST rank's time; both ranks agree on every reduced number
    assert l0["value"] == pytest.approx(2 * (1 << 20) * 2000 / 0.084) and l0["value"] == l1["value"]
    assert l0["roofline"]["kernel_us"] == pytest.approx(40.2) and l0["rccl"]["kernel_us_min"] == pytest.approx(38.0)
    assert l0["rccl"]["rccl_ranks"] == 2 and l0["rccl"]["backend"] == "gloo"
    assert l0["rccl"]["with_rollout_gather"]["episodes_seen"] == 3.0
    assert l0["roofline"]["traffic"] is None and l0["roofline"]["traffic_source"] is None


def test_valu_object_from_the_committed_profile():
    """bench.py's `valu` object (the step kernel's second resource): the launch's vector instructions (committed PMC pass of the
    workload) priced at the MEASURED issue rates -- 1.2 / 1.8 / 3.4 ns, not the quad-cycle the SQ counters tick in (round-3 review,
    item 4) -- only for the profiled workload, and labelled as a static file."""
    import bench

    t = bench.valu_issue_time_us(15.57e6, 0.81e6)
    assert 18.0 < t["floor_us"] < 18.5 and t["floor_us"] < t["priced_us"] < 22.0        # 950 instructions x 16 tiles per SIMD
    v = bench.valu_from_profile("raw", "rk4", 1 << 20, 1, 0.0367)
    assert v is not None and 0.45 < v["share_of_launch"] < 0.70 and "static file" in v["source"]
    assert v["issue_time_floor_us_per_simd"] < v["issue_time_us_per_simd"] < 25.0
    assert 900 < v["instructions_per_64_env_tile"] < 1000
    assert "busy_frac" not in v, "the 4-cycles-per-instruction figure is gone from the line"
    assert bench.valu_from_profile("raw", "euler", 1 << 20, 1, 0.0375) is None
    assert bench.valu_from_profile("raw", "rk4", 1 << 16, 1, 0.0375) is None
    assert bench.valu_from_profile("raw", "rk4", 1 << 20, 16, 0.0375) is None
    b = bench.small_batch_bound(65536, "euler", 0.04, 16, 38e-6)
    assert b["bound"] == "dependent_chain_of_one_tile_per_simd" and 0.1 < b["frac"] < 0.6 and "share_of_wave_cycles" in b


def test_rollout_gather_loop_is_warmed_and_itemised(monkeypatch):
    """bench.py's N > 1 second loop (BASELINE configs[4]'s per-rollout all-gather): the first gather is timed alone, one
    untimed pass with two gathers warms the path, then the same steps run without and with gathers on the same env and the
    record carries the difference per gather.  GPU-free: the timed loop and the env are stand-ins."""
    import types

    import torch

    calls = []

    class FakeEnv:
        def episode_stats_tensor(self):
            return torch.arange(8, dtype=torch.float64)

    def fake_timed_loop(env, pool, steps, k, barrier, stream, torch_mod, every=None):
        n_cb = 0
        if every is not None:
            for j in range(steps):
                if (j + 1) % every[0] == 0:
                    every[1]()
                    n_cb += 1
        calls.append((steps, n_cb))
        return steps * 45e-6 + n_cb * 30e-6, 0.045

    monkeypatch.setattr(bench, "timed_loop", fake_timed_loop)
    from rl_rocket_6dof_amd import dist as r6dist

    args = bench.parse_args(["--gpus", "2", "--steps", "2000"])
    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(synchronize=lambda dev=None: None))
    g = bench.rollout_gather_loop(FakeEnv(), [None, None], args, lambda: None, None, fake_torch, r6dist, None)
    assert calls == [(64, 2), (320, 0), (320, 10)]                     # warm pass with 2 gathers, then plain, then with gathers
    for key in ("gather_every", "steps", "gathers", "wall_s", "us_per_step_tracked_env", "us_per_step_with_gathers", "us_per_gather",
                "first_gather_us", "warm", "episodes_seen"):
        assert key in g, key
    assert g["gathers"] == 10 and g["steps"] == 320
    assert g["us_per_gather"] == pytest.approx(30.0, rel=1e-6) and g["us_per_step_tracked_env"] == pytest.approx(45.0, rel=1e-6)
    assert g["first_gather_us"] >= 0.0


def test_all_ranks_agree_without_a_process_group():
    from rl_rocket_6dof_amd import dist as r6dist

    assert r6dist.all_ranks_agree(True, "k") is True and r6dist.all_ranks_agree(False, "k") is False


def test_cpu_baseline_all_cores_reports_cores_and_model():
    r = bench.cpu_baseline_all_cores(1.0, max_procs=2)
    assert "error" not in r, r
    assert r["cores"] == min(2, bench.usable_cores()) and r["value"] > 0 and r["cpu_model"] and "independent processes" in r["sample"]
