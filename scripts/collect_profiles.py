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
if os.path.isdir(pmc):
    res = collections.OrderedDict()
    newest = {}      # gpurun_out/ keeps the files of earlier passes next to the latest: one (the newest) per counter group
    for f in glob.glob(os.path.join(pmc, "g*", "**", "*counter_collection.csv"), recursive=True):
        g = os.path.relpath(f, pmc).split(os.sep)[0]
        if g not in newest or os.path.getmtime(f) > os.path.getmtime(newest[g]):
            newest[g] = f
    for f in (newest[g] for g in sorted(newest)):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r["Kernel_Name"]:
                by[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in by.items():
            v = v[-30:]
            res[k] = sum(v) / len(v)
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
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_valu.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_pmc_traffic.json")).read())
