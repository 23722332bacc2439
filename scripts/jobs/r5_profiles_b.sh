# round 5, part B: the GPU test suite, smoke, the default bench command and the driver's command shape, untraced
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputest_2.log 2>&1 || { tail -40 gpurun_out/r5_gputest_2.log; exit 1; }
tail -1 gpurun_out/r5_gputest_2.log
python -c "import __graft_entry__ as g; g.smoke()"
python3 bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_shape.json 2> gpurun_out/r5_bench_driver_shape.err
python3 - <<'PY'
import json
for f in ("gpurun_out/r5_bench_default.json", "gpurun_out/r5_bench_driver_shape.json"):
    b = json.load(open(f))
    print(f, "ms_per_step", round(b["ms_per_step"] * 1e3, 2), "kernel_us", round(b["roofline"]["kernel_us"], 2), "frac", round(b["roofline"]["frac"], 3),
          "steady median", round(b["steady"]["median"], 2), "p10", round(b["steady"]["p10"], 2), "p90", round(b["steady"]["p90"], 2),
          "| rk45", round(b["rk45"]["us_per_env_step"], 1), "k16", "%.3g" % b["configs1"]["k16"]["value"], "1-launch euler", round(b["configs1"]["one_launch_per_step"]["us_per_env_step"], 2),
          "ooc", round(b["out_of_cache"]["us_per_env_step"], 1), "two_groups gain", round(b["two_groups"].get("gain_us", 0), 2),
          "ppo", "%.3g" % b["ppo_training"]["env_steps_per_s"], "configs4", "%.3g" % b["ppo_configs4"].get("env_steps_per_s", 0), b["ppo_configs4"].get("error"))
PY
