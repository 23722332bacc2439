#!/usr/bin/env python3
"""Copies the judged summaries of a profiling round from gpurun_out/ (scratch) into profiles/ (tracked):
    gpurun_out/profile_<tag>/summary.json, trace/**/kernel_stats.csv   <- scripts/profile_round.sh <tag>
    gpurun_out/pmc_<tag>/g*/**/counter_collection.csv                  <- scripts/pmc.sh <tag>
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
src, pmc, dst = (os.path.join(ROOT, "gpurun_out", f"profile_{tag}"), os.path.join(ROOT, "gpurun_out", f"pmc_{tag}"),
                 os.path.join(ROOT, "profiles"))
summ = json.load(open(os.path.join(src, "summary.json")))
json.dump(summ, open(os.path.join(dst, f"{tag}_rocprofv3_summary.json"), "w"), indent=1)
ks = max(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # gpurun_out/ keeps earlier runs: the newest one
shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_under_trace.log")) if l.startswith("{")]
if line:
    json.dump(json.loads(line[-1]), open(os.path.join(dst, f"{tag}_bench_line_under_trace.json"), "w"), indent=1)
traffic = {k: summ[k] for k in ("FETCH_SIZE_KB_per_launch_raw", "WRITE_SIZE_KB_per_launch_raw", "hbm_bytes_per_launch",
                                "hbm_read_bytes_per_launch_corrected", "hbm_write_bytes_per_launch", "algorithmic_bytes_per_launch", "commit")}
traffic["method"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python3 bench.py --steps 50 "
                     "--no-cpu-baseline --no-extras`; mean over the last 50 r6::step_kernel launches; FETCH_SIZE doubled (gfx950 "
                     "counts 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section), WRITE_SIZE as is; KB = 1024 B. The "
                     "counters sit on the fabric side of L2 and include Infinity-Cache hits.")
traffic["ratio_to_algorithmic"] = traffic["hbm_bytes_per_launch"] / traffic["algorithmic_bytes_per_launch"]
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
if os.path.exists(os.path.join(pmc, "summary.json")):
    res = collections.OrderedDict(json.load(open(os.path.join(pmc, "summary.json"))))     # scripts/pmc.sh: mean per launch of every counter
    tiles = (1 << 20) / 64
    cyc = res.get("GRBM_GUI_ACTIVE", 0) / 8
    out = {"command": "scripts/pmc.sh (rocprofv3 --kernel-trace --pmc <group>, one counter group per run, python3 bench.py --steps 30 "
                      "--warmup 300 --no-cpu-baseline --no-extras); mean over the last 30 r6::step_kernel<1,0,14,false,true> launches",
           "commit": summ.get("commit"), "raw_per_launch": res,
           "derived": {"valu_instructions_per_64_env_tile": res.get("SQ_INSTS_VALU", 0) / tiles,
                       "salu_instructions_per_tile": res.get("SQ_INSTS_SALU", 0) / tiles,
                       "kernel_cycles_under_pmc (GRBM_GUI_ACTIVE / 8 XCDs)": cyc,
                       "valu_busy_fraction (4 * SQ_ACTIVE_INST_VALU / 1024 SIMDs / kernel cycles)": 4 * res.get("SQ_ACTIVE_INST_VALU", 0) / 1024 / cyc if cyc else None,
                       "mean_resident_waves_per_simd (SQ_WAVE_CYCLES * 4 / 1024 / kernel cycles)": 4 * res.get("SQ_WAVE_CYCLES", 0) / 1024 / cyc if cyc else None}}
    # the VALU work priced at the MEASURED issue rates (bench.py valu_issue_time_us), not at the quad-cycle SQ_ACTIVE_INST_VALU ticks in
    sys.path.insert(0, ROOT)
    import bench
    insts, trans = res.get("SQ_INSTS_VALU", 0.0), res.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    t = bench.valu_issue_time_us(insts, trans)
    out["derived"]["transcendental_instructions_per_tile"] = trans / tiles
    out["derived"]["valu_issue_time_us_per_simd (measured rates: 1.2 ns plain, 1.8 ns scalar source x assumed share, 3.4 ns transcendental)"] = t["priced_us"]
    out["derived"]["valu_issue_time_floor_us_per_simd (all at 1.2 ns)"] = t["floor_us"]
    wc = res.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        out["derived"]["share_of_wave_cycles"] = {"issuing (SQ_ACTIVE_INST_ANY)": res.get("SQ_ACTIVE_INST_ANY", 0) / wc, "waiting in s_waitcnt (SQ_WAIT_ANY)": res.get("SQ_WAIT_ANY", 0) / wc,
                                                   "issue stall (SQ_WAIT_INST_ANY)": res.get("SQ_WAIT_INST_ANY", 0) / wc}
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_valu.json"), "w"), indent=1)
stepk = os.path.join(ROOT, "gpurun_out", f"pmc_stepk_{tag}", "summary.json")
if os.path.exists(stepk):
    raw = json.load(open(stepk))
    raw.pop("driver_line", None)
    g1 = os.path.join(ROOT, "gpurun_out", f"pmc_stepk_{tag}", "g1.log")
    line = next((l.strip() for l in reversed(open(g1, errors="replace").read().split("\n")) if "env-steps/s" in l), None) if os.path.exists(g1) else None
    if line:
        line += "   (wall time per launch UNDER counter collection, which serialises launches: not a throughput figure)"
    wc, tiles_k, K = raw.get("SQ_WAVE_CYCLES", 0.0), 1024.0, 16.0
    d = {"waves_per_launch": raw.get("SQ_WAVES"), "waves_per_simd": (raw.get("SQ_WAVES") or 0) / 1024.0,
         "valu_instructions_per_tile_and_env_step": raw.get("SQ_INSTS_VALU", 0) / tiles_k / K,
         "salu_instructions_per_tile_and_env_step": raw.get("SQ_INSTS_SALU", 0) / tiles_k / K,
         "lds_instructions_per_tile_and_env_step": raw.get("SQ_INSTS_LDS", 0) / tiles_k / K}
    if wc:
        d["share_of_wave_cycles"] = {"valu_issue": raw.get("SQ_ACTIVE_INST_VALU", 0) / wc, "scalar_issue": raw.get("SQ_ACTIVE_INST_SCA", 0) / wc,
                                     "lds_issue": raw.get("SQ_ACTIVE_INST_LDS", 0) / wc, "waiting_s_waitcnt_or_sleep": raw.get("SQ_WAIT_ANY", 0) / wc,
                                     "issue_stall": raw.get("SQ_WAIT_INST_ANY", 0) / wc}
    json.dump({"kernel": "r6::step_k_pair_kernel<0,0,14> (Euler, acceleration shaping, 14-wide obs): four waves per 64-env tile -- one for the dynamics / reset, "
                         "two for the reward of alternate steps, one for the per-step stores -- on one SIMD",
               "workload": "BASELINE configs[1]: 65 536 envs, Euler dt 0.04, r6_step_k with K = 16 steps per launch, all per-step outputs stored (scripts/stepk_driver.py)",
               "method": "scripts/pmc_stepk.sh: rocprofv3 --kernel-trace --pmc <group>, one counter group per run, mean over the last 30 launches",
               "commit": summ.get("commit"), "driver_line": line, "raw_per_launch": raw, "derived": d},
              open(os.path.join(dst, f"{tag}_pmc_stepk.json"), "w"), indent=1)
rk45 = os.path.join(ROOT, "gpurun_out", f"pmc_rk45_{tag}", "summary.json")
if os.path.exists(rk45):
    raw = json.load(open(rk45))
    json.dump({"kernel": "r6::step_kernel<2,0,14,false,false> (in-kernel adaptive RK45, scipy's step control), 2^20 envs, steady state of the bench workload",
               "method": "scripts/pmc_rk45.sh: rocprofv3 --kernel-trace --pmc <group>, one counter group per run of scripts/rk45_driver.py, mean over the last 30 launches; "
                         "defer_0 = every wave follows its slowest lane (r6_tune rk45_defer 0), defer_6 = the default: a wave's last <= 6 unfinished lanes become guests",
               "commit": summ.get("commit"), **raw}, open(os.path.join(dst, f"{tag}_pmc_rk45.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_pmc_traffic.json")).read())
