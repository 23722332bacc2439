#!/usr/bin/env python3
"""profiles/r5_experiments.json: the A/B records and probes of round 5, from the logs under gpurun_out/ (scratch) -- per experiment the
commit the code is at, the build command of every variant library, the measured figures.  python scripts/collect_experiments_r5.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
rd = lambda f: open(os.path.join(G, f), errors="replace").read() if os.path.exists(os.path.join(G, f)) else ""
js = lambda f: json.load(open(os.path.join(G, f))) if os.path.exists(os.path.join(G, f)) else None


def pooled(f):
    out = {}
    for ln in rd(f).split("\n"):
        m = re.match(r"POOLED (\S+)\s+us/step median\s+([\d.]+) \(([+-][\d.]+) vs (\S+)\) lower-half mean\s+([\d.]+) min\s+([\d.]+)\s+frac ([\d.]+)", ln)
        if m:
            out[m.group(1)] = {"median_us": float(m.group(2)), "delta_vs_first_us": float(m.group(3)), "lower_half_mean_us": float(m.group(5)), "min_us": float(m.group(6)), "hbm_frac": float(m.group(7))}
    return out


def kbench(f):
    out, lib = {}, None
    for ln in rd(f).split("\n"):
        if ln.strip().endswith(".so"):
            lib = os.path.basename(ln.strip())
            out.setdefault(lib + (" (again)" if lib in out else ""), {})
            lib = list(out)[-1]
        m = re.match(r"\s+K=\s*(\d+): all outputs\s+([\d.]+) us/step.*last-only\s+([\d.]+) us/step", ln)
        if m:
            out.setdefault(lib or "libr6dof.so", {})["K=" + m.group(1)] = {"all_outputs_us_per_step": float(m.group(2)), "last_only_us_per_step": float(m.group(3))}
    return out


E = {}
E["_method"] = ("scripts/ab_swap.py: ONE env whose library handle / r6_tune shape is swapped between interleaved timing rounds (every build steps the same memory); "
                "POOLED = over the envs of a session.  Variant libraries: scripts/build_variant.sh <name> <-D switches> (only r6_kernels.hip is recompiled; every switch "
                "used this round lives in r6_kernels.hip / r6_device.hpp's functions instantiated there).  Boxes differ by +-3 % (and this round by up to 6 %) in absolute level: "
                "compare within a session only.")
p1 = js("r5_driver_shape_probe.json")
tr = js("r5_trace_driver_shape_summary.json")
E["driver_shape (review round 4, Next 1)"] = {
    "commit": "748eef2 (library of round 4, 7a7c8a2+dirty: only hidden symbols differ)",
    "what": "the driver's command `python3 bench.py --gpus 1 --steps 20 --warmup 5` on a fresh box, four times; then the per-launch picture (scripts/driver_shape_probe.py) and the kernel trace of the same command",
    "bench_runs_us_per_step (wall | HIP events)": [[round(js(f"r5_driver_shape_{i}.json")["ms_per_step"] * 1e3, 2), round(js(f"r5_driver_shape_{i}.json")["roofline"]["kernel_us"], 2)] for i in (1, 2, 3, 4) if js(f"r5_driver_shape_{i}.json")],
    "note_runs": "run 1 = the first process on the box, with the side figures; runs 2-4 --no-extras --no-cpu-baseline",
    "idle_gap_before_a_20_launch_block": [{"preroll_s": b["preroll_s"], "idle_gap_s": b["idle_gap_s"], "bracket_event_us_sorted": b["bracket_event_us_sorted"], "first_launch_us": b["first_launch_us"]} for b in (p1 or {}).get("blocks", [])],
    "steady_20_launch_chunk_means_us": (p1 or {}).get("steady_20_launch_chunk_means_us"),
    "trace_of_the_command": tr and {"warmup5_and_timed20_durations_us": [round(r["dur_us"], 1) for r in tr["warmup5_and_timed20"]], "gap_between_warmup_and_timed_us": round(max(tr["gaps_us"]), 1),
                                     "timed20_mean_kernel_us": tr["timed20_mean_dur_us"], "burnin_mean_us_by_200_launches": [round(x, 1) for x in tr["burnin_mean_us_by_200"]]},
    "reading": "a 20-launch region pays +0.8 us per step for the first launch after the barrier (HIP events) and +0.9 us more for the host's wake-up (wall clock); after >= 20 ms of idleness the whole "
               "block runs at 39.4-45 us (the chip has clocked down); the first 30 ms of a process run 32.7 -> 53 -> 36 us per launch"}
def _ds(f):
    b = js(f)
    return b and {"file": f, "lib": b["config"]["lib_commit"], "wall_us": round(b["ms_per_step"] * 1e3, 2), "hip_events_us": round(b["roofline"]["kernel_us"], 2), "frac": round(b["roofline"]["frac"], 3),
                  "steady_median_us": round(b["steady"]["median"], 2), "steady_p90_us": round(b["steady"]["p90"], 2), "timed_region_over_median": round(b["steady"]["timed_region_over_median"], 3)}
E["driver_shape (review round 4, Next 1)"]["untraced_runs_with_the_round_5_bench (1 s pre-roll, steady record)"] = [x for x in (
    {"file": "profiles/r5_bench_line_driver_shape_slow_episode.json", "lib": "ce05765", "wall_us": 42.0, "hip_events_us": 41.0, "frac": 0.604, "steady_median_us": 36.97, "steady_p90_us": 39.99, "timed_region_over_median": 1.109},
    {"file": "r5_bench_driver_shape_final.json (as first taken; the file was later overwritten by the run below)", "lib": "d6dc237", "wall_us": 37.17, "hip_events_us": 36.21, "frac": 0.684,
     "steady_median_us": 36.88, "steady_p90_us": 41.12, "timed_region_over_median": 0.982},
    _ds("r5_bench_driver_shape.json"), _ds("r5_bench_driver_shape_final.json"), _ds("r5_bench_driver_shape2.json")) if x]
E["driver_shape (review round 4, Next 1)"]["untraced_runs_note"] = "three fresh boxes: two of the three 20-launch samples sat in a clock dip (1.09-1.11 x their own steady median), one did not (0.99)"
tp = js("r5_timed_region_probe.json")
if tp:
    E["driver_shape (review round 4, Next 1)"]["a_20_launch_region_behind_a_barrier_against_one_in_a_continuous_stream (scripts/timed_region_probe.py, 40 interleaved repetitions, HIP events, us per launch)"] = {
        "A: bench.py's sequence ([200 launches, sync] x 3, barrier, 5 warm-up, barrier, barrier, 20 timed)": {k: v for k, v in tp["A"].items() if k != "values"},
        "B: no synchronize between pre-roll and warm-up, ONE barrier before the 20 timed": {k: v for k, v in tp["B"].items() if k != "values"},
        "C: the 20 launches inside a continuous stream (no synchronize before them)": {k: v for k, v in tp["C"].items() if k != "values"},
        "reading": "the barrier the contract puts in front of the timed region costs a 20-launch sample ~1.5 us per launch (4 %) against the same launches in a continuous stream -- 0.8 us of it the first launch's start-up -- "
                   "and doubles its chance of reading > 1.06 x the continuous median (0.35-0.45 against 0.17); the additional synchronizes of bench.py's sequence change nothing (A = B)"}
b1, b2 = js("r5_burst_probe.json"), js("r5_burst_probe2.json")
if b1 and b2:
    import statistics as st
    q = lambda v, f: sorted(v)[int(f * len(v))]
    E["slow_episodes (what the 20 launches of BENCH_r04 hit)"] = {
        "commit": "748eef2", "scripts": "scripts/burst_probe.py, scripts/burst_probe2.py, scripts/clock_per_launch.sh",
        "chunk_means_of_20_launches_us": {k: {"median": q(b1[k]["us"][1:], 0.5), "p10": q(b1[k]["us"][1:], 0.1), "p90": q(b1[k]["us"][1:], 0.9)} for k in ("step", "frozen_population_no_reset", "membench_pattern") if "us" in b1.get(k, {})},
        "episode_end_share_per_step": {"mean": st.mean(b1["step"]["done_share"]), "std": st.pstdev(b1["step"]["done_share"])},
        "same_deterministic_sequence_twice": {"corr_of_chunk_means": b2["corr_A_B"], "event_counts_identical": b2["identical_event_counts"],
                                               "slow_chunks_A_B_common (> 1.06 x median)": [sum(1 for x in b2["A"]["us"][1:] if x > 1.06 * q(b2["A"]["us"][1:], 0.5)), sum(1 for x in b2["B"]["us"][1:] if x > 1.06 * q(b2["B"]["us"][1:], 0.5)),
                                                                                            sum(1 for x, y in zip(b2["A"]["us"][1:], b2["B"]["us"][1:]) if x > 1.06 * q(b2["A"]["us"][1:], 0.5) and y > 1.06 * q(b2["B"]["us"][1:], 0.5))]},
        "launches_serialised_by_counter_collection (idle gaps between them)": (js("clock_r5/summary.json") or {}).get("duration_us"),
        "reading": "slow episodes of 1.5-3 ms at +5 ... +15 % arrive every 8-25 ms; they do not replay with the data, are absent from the constant-work launches of the same kernel "
                   "(no episode ends: 33.8 us, +-0.8 %) and from the memory skeleton (31.1 us, +-0.3 %), and absent when the launches are spaced out (35.9 us, p10 35.3, p90 36.8): the chip's clock management "
                   "under sustained full load (round 4's clock probe: 2.03-2.19 GHz during RK4 launches against 2.38 GHz for the skeleton), not the workload"}
cv = js("r5_clock_vs_duration.json")
if cv:
    import numpy as np
    us, g = np.array(cv["series_us"][1:]), np.array(cv["series_ghz"][1:])     # the first chunk carries the set-up of the copy path
    med = float(np.median(us)); slow = us > 1.04 * med
    E["shader_clock_against_chunk_time (back to back; -DR6_CLOCK_PROBE -DR6_CLOCK_RING=10 build at commit after 96c78ca, scripts/clock_vs_duration.py)"] = {
        "what": "199 chunks of 20 launches of the bench workload right after a synchronize + allocation; every workgroup of every launch writes its s_memtime cycles over its 100 MHz ticks into a ring the host copies every 10 launches",
        "corr(chunk time, chunk shader clock)": float(np.corrcoef(us, g)[0, 1]), "chunk_time_us": {"median": med, "p10": float(np.percentile(us, 10)), "p90": float(np.percentile(us, 90))},
        "clock_ghz": {"median": float(np.median(g)), "min": float(g.min()), "max": float(g.max())},
        "chunks_above_1.04_x_median": {"n": int(slow.sum()), "clock_ghz": float(g[slow].mean()), "time_us": float(us[slow].mean())}, "the_others": {"clock_ghz": float(g[~slow].mean()), "time_us": float(us[~slow].mean())},
        "first_60_chunks_us": [round(float(x), 1) for x in us[:60]], "first_60_chunks_ghz": [round(float(x), 2) for x in g[:60]],
        "reading": "the launch time follows the shader clock (correlation -0.91): 54 us at 1.56 GHz right after the idle gap, 38.5 us at 2.35 GHz, the dips inside the steady state (40.2-40.3 us) at 2.12-2.14 GHz; "
                   "(with a copy after EVERY launch the launches are spaced out and the dips do not occur: 40.7 us, p10 40.3, p90 41.7 including the copy)"}
E["packed_fp32 (review round 4, Next 3)"] = {"commit": "9cdac53", "script": "scripts/pk_rate_bench.hip", "ns_per_wave64_instruction_and_simd": [l for l in rd("r5_pk_rate_bench.log").split("\n") if l.startswith("waves/SIMD")],
                                             "reading": "a v_pk_fma_f32 (two float operations per lane) costs 2.15-2.24 ns at four / five waves per SIMD against 1.14-1.22 ns for a plain v_fma_f32 with vector sources: two plain "
                                                        "instructions are 2.3-2.4 ns -- packing saves at most 7 % on the PACKABLE instructions if no register move is added, i.e. < 3 % of an RK45 attempt: dropped by the review's own stop rule, "
                                                        "on the issue cost rather than the instruction count.  (One wave per SIMD: 2.72 against 2 x 2.35 ns.)"}
E["dependent_chains_and_transcendentals"] = {"commit": "8b63f76", "script": "scripts/dep_chain_bench.hip", "ns_per_instruction": [l for l in rd("r5_dep_chain_bench.log").split("\n") if l.startswith("waves/SIMD")],
                                             "reading": "from three waves per SIMD on even a single dependent chain per wave issues at 1.1-1.3 ns per instruction; a v_rcp_f32 among seven v_fma costs 3.7-8.3 ns"}
E["two_groups (review round 4, Next 4)"] = {"commit": "9cdac53", "code": "BatchedRocket6DOF(groups=2, group_sync='none'), bench.two_groups_side_measurement", "first_box": js("r5_two_groups.json") and {k: v for k, v in js("r5_two_groups.json").items() if k != "workload"},
                                            "reading": "gain 0.70 us on the first box, 1.44 us in the traced bench run of profiles/r5_bench_line_under_trace.json: below / at the review's 1.5 us; the API stays (tested bit-identical), the figure stays in the line as a labelled side figure"}
E["rk45 (review round 4, Next 2a)"] = {
    "attempts_in_the_bench_steady_state": (js("r5_rk45_attempts.json") or {}).get("rep2"),
    "1_workgroup_stash (commit 9cdac53: the form the review asked for)": pooled("r5_swap_rk45_defer.log"),
    "1b_workgroup_stash_pmc (commit 8b63f76 is the last with it)": js("pmc_rk45_r5/summary.json") and "superseded by profiles/r5_pmc_rk45.json (guests); the stash's own pass: VALU per tile 4107 -> 3437, SQ_WAIT_ANY 72.9M -> 94.2M, duration under counters 121.0 -> 113.7 us (gpurun_out/r5_call6.log)",
    "2_workgroup_dynamic_tile_hand_out (build_variant.sh wgdyn -DR6_RK45_WG_DYNAMIC=1, commit 8b63f76)": pooled("r5_swap_rk45_wgdyn.log"),
    "3_register_diet (commit 8b63f76)": pooled("r5_swap_rk45_diet.log") and "defer=0 114.3 -> 111.1 us, with the workgroup stash 107.1 (gpurun_out/r5_call7.log)",
    "4_guests_thresholds (commit 27c84bb)": pooled("r5_swap_rk45_guests.log"),
    "5_guests_with_list_recycling (commit 57fde78; another box)": pooled("r5_swap_rk45_recycle.log"),
    "6_other_sizes (commit 57fde78; shapes: default grid | grid pinned to 1024 workgroups)": {n: pooled(f"r5_swap_rk45_grid_{n}.log") for n in (2097152, 4194304, 8388608)},
    "7_tiles_per_wave_of_the_balanced_grid_at_4M_envs": pooled("r5_swap_rk45_tpw.log")}
E["step_k (review round 4, Next 2b)"] = {
    "trace_before (commit 57fde78, -DR6_TRACE_K build)": [l[:400] for l in rd("r5_stepk_pair_trace.log").split("\n") if "rep 2" in l],
    "trace_after_per_role_kernarg (commit af3c13a)": [l[:400] for l in rd("r5_stepk_pair_trace2.log").split("\n") if "rep 2" in l],
    "wave_priorities (build_variant.sh c3 / c3a1 / c3b2a1 / b2: -DR6_PAIR_PRIO_A/B/C; commit before af3c13a)": kbench("r5_call16.log"),
    "wave_priorities_with_the_reset_draw_raised (dyn1-4: + -DR6_PAIR_PRIO_RESET=3)": kbench("r5_call17.log"),
    "steps (us per env step, 65 536 envs, all outputs; scripts/kstep_bench.py)": {"round 4": {"K=4": 4.18, "K=16": 2.45, "K=64": 1.90}, "reward waves' episode-return load in front of the K loop": {"K=4": 4.21, "K=16": 2.43, "K=64": 1.88},
                                                                                  "+ every role re-reads its kernel arguments": {"K=4": 4.11, "K=16": 2.38, "K=64": 1.66}, "+ reset draw from LDS, converted in its lanes": {"K=4": 3.91, "K=16": 2.27, "K=64": 1.61},
                                                                                  "+ the dynamics wave takes the last step's reward (-DR6_PAIR_A_LAST=1, not kept)": {"K=4": 4.02, "K=16": 2.29, "K=64": 1.58}}}
E["reset_drain_one_env_per_quad (commit af3c13a; build_variant.sh noquad -DR6_RESET_QUAD=0)"] = {"rk4_raw": pooled("r5_swap_quad_rk4.log"), "euler": pooled("r5_swap_quad_euler.log"), "fused_stack": pooled("r5_swap_quad_wrapped.log")}
E["this_round_against_round_4_on_one_set_of_buffers (build: round 4's sources at 576a206 compiled with the same flags -> variants/libr6dof_r4.so; library af3c13a+: quad reset drain still in the RK4 instances)"] = {
    "rk4_raw": pooled("r5_swap_r4_vs_r5_rk4.log"), "euler": pooled("r5_swap_r4_vs_r5_euler.log"), "fused_stack": pooled("r5_swap_r4_vs_r5_wrapped.log"), "rk45": pooled("r5_swap_r4_vs_r5_rk45.log")}
E["grid_barrier_vs_kernel_boundary (review round 4, Next 6)"] = {"commit": "af3c13a", "script": "scripts/grid_barrier_bench.hip", "result": rd("r5_grid_barrier_bench.log").strip()}
E["ppo_all_reduce_in_the_epoch_graph (review round 4, Next 5)"] = js("ppo_rccl_graph.json")
def _abtab(f):
    out = {}
    for ln in rd(f).split("\n"):
        m = re.match(r"(\S+)\s+n=(\d+) rows=\d+ (.+?): gradient \+ reduce median\s+([\d.]+) us\s+min\s+([\d.]+) us", ln)
        if m:
            out.setdefault(m.group(3), {})[m.group(1)] = {"median_us": float(m.group(4)), "min_us": float(m.group(5))}
    return out
E["rk4_arithmetic_share (library dcb3613; scripts/jobs/r5_job33.sh: t_go's Newton loop cut to 6 / 4 iterations behind a throw-away switch, -DR6_ABLATE_REWARD; timing only)"] = pooled("r5_swap_tgo.log")
E["ppo_gradient_launch_levers (library dcb3613; scripts/jobs/r5_job34.sh -DR6U_ABLATE=2 / =7, r5_job35.sh -fslp-vectorize on every unit; scripts/ppo_grad_ab.py)"] = {
    "products_left_out (timing only, wrong gradients)": _abtab("r5_ppo_ablate.log"), "slp_vectoriser_on": _abtab("r5_ppo_slp.log")}
E["mfma_results_in_vector_registers (-mllvm -amdgpu-mfma-vgpr-form; scripts/jobs/r5_job37.sh, r5_job38.sh; adopted in build.py at d6dc237)"] = {
    "gradient_plus_reduce": _abtab("r5_ppo_vform.log"), "rollout_kernel_two_processes_per_library": [ln for ln in rd("r5_rollout_vform.log").split("\n") if ln.startswith("==") or "fused r6_rollout" in ln]}
E["dw2_products_straight_into_the_launch_sums (-DR6U_DW2_DIRECT=1, off by default; scripts/jobs/r5_job39.sh - r5_job42.sh)"] = {
    "timing_build_with_sums_running_to_infinity (misleading: the chip clocks higher on such data)": _abtab("r5_ppo_direct.log"),
    "the_exact_form (direct = shift 16, sh10 = first tile's largest entry at 2^3, nodirect = default)": _abtab("r5_ppo_direct_ab.log"),
    "placement_of_the_first_tile (sh0 .. sh16)": _abtab("r5_ppo_shift.log"), "with_and_without_MODE.FP16_OVFL": _abtab("r5_ppo_ovfl.log"),
    "accuracy_against_float64_autograd": [ln for ln in rd("r5_ppo_direct_acc.log").split("\n") if "autograd" in ln]}
_pp = [ln for ln in rd("r5_preroll_probe.log").split("\n") if ln.startswith("{")]
if _pp:
    E["preroll_length_against_the_20_launch_sample (scripts/preroll_probe.py, scripts/jobs/r5_job43.sh; library fa2a3e3)"] = {
        "what": "from an idle chip: P seconds of stepping as bench.py's pre-roll does, 5 warm-up launches, barrier, 20 launches between two events, then 1000 launches for the steady median of the same moment; 8 trials per P in random order",
        "by_preroll_seconds": json.loads(_pp[-1]), "reading": "9 of 48 samples above 1.06 x their own steady median (19 %: the share of 20-launch windows a continuous run has in slow episodes); 0 of 8 at 1 s, 3 of 8 at 4 s"}
_pd = [ln for ln in rd("r5_preroll_duty.log").split("\n") if ln.startswith("{")]
if _pd:
    E["preroll_duty_cycle_against_the_20_launch_sample (scripts/preroll_duty_probe.py, scripts/jobs/r5_job44.sh)"] = {
        "modes": "full = batches of 200 launches between synchronizes (bench.py until 2970fd0), sync4 / sync1 = a synchronize every 4 / every launch, gap = + 30 us of host sleep",
        "by_mode": json.loads(_pd[-1]), "trials": [ln for ln in rd("r5_preroll_duty.log").split("\n") if "launches in 1 s" in ln]}
if rd("r5_driver_shape_b4.log"):
    E["driver_shape_with_the_batch_4_preroll (scripts/jobs/r5_job45.sh; eight + four fresh processes on one box, then the default command)"] = rd("r5_driver_shape_b4.log").strip().split("\n")
E["compiler_scheduling_strategies_for_the_step_kernels (scripts/jobs/r5_job46.sh, r5_job47.sh; none kept)"] = {
    "rk4_raw_first_session (max-ilp, amdgpu-trackers, max-memory-clause)": pooled("r5_swap_sched.log"),
    "amdgpu_trackers": {"rk4_raw": pooled("r5_swap_trk_rk4.log"), "euler": pooled("r5_swap_trk_euler.log"), "fused_stack": pooled("r5_swap_trk_wrapped.log"),
                        "rk45": pooled("r5_swap_trk_rk45.log"), "rk4_2^23_envs": pooled("r5_swap_trk_2p23.log"), "step_k (base, trackers, base, trackers)": kbench("r5_kstep_trk.log")}}
E["compiler_code_generation_switches_for_the_step_kernels (scripts/jobs/r5_job51.sh; library 0a21da9 sources; none kept)"] = {
    "build": "scripts/build_variant.sh <name> -mllvm <switch>: prealloc = -amdgpu-prealloc-sgpr-spill-vgprs, nopostra = -enable-post-misched=0, relaxed = "
             "-amdgpu-schedule-relaxed-occupancy, greedyrc = -greedy-regclass-priority-trumps-globalness, setprio = -amdgpu-set-wave-priority; "
             "-amdgpu-max-memory-clause=32 and -amdgpu-schedule-metric-bias=100 give byte-identical step kernels (disassembly compared)",
    "rk4_raw": pooled("r5_swap_cg_rk4.log"), "euler_dt_0.04": pooled("r5_swap_cg_euler.log"), "rk45": pooled("r5_swap_cg_rk45.log")}
if js("r5_single_env_rate.json"):
    E["host_driven_stepping_through_one_pinned_image (scripts/single_env_rate.py, scripts/jobs/r5_job50.sh)"] = js("r5_single_env_rate.json")
if js("r5_single_env_rate_poll.json"):
    E["host_driven_stepping_with_a_polling_runtime (HSA_ENABLE_INTERRUPT=0, scripts/jobs/r5_job53.sh; same box, interrupt | polling: no gain)"] = {
        "interrupt": js("r5_single_env_rate_irq.json"), "polling": js("r5_single_env_rate_poll.json")}
_c4 = rd("r5_ppo_cfg4_mb.log")
if "{" in _c4:
    E["ppo_configs4_minibatch_size (scripts/jobs/r5_job36.sh)"] = json.loads(_c4[_c4.index("{"):])
json.dump(E, open(os.path.join(ROOT, "profiles", "r5_experiments.json"), "w"), indent=1)
print("wrote profiles/r5_experiments.json:", list(E))
