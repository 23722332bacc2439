#!/usr/bin/env python3
"""Static instruction mix of the step kernels: hipcc -S then count per kernel (dev tool)."""
import re, subprocess, sys, os, collections
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl_rocket_6dof_amd/csrc/r6_kernels.hip")
out = "/tmp/r6s/r6.s"
os.makedirs("/tmp/r6s", exist_ok=True)
subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-math-errno", "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
cur = None; stats = {}
for l in lines:
    m = re.match(r"^(_ZN2r6\w+):", l)
    if m: cur = m.group(1); stats[cur] = collections.Counter(); continue
    if l.startswith(".Lfunc_end"): cur = None; continue
    if cur is None: continue
    t = l.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"): continue
    stats[cur][t.split()[0]] += 1
want = sys.argv[1] if len(sys.argv) > 1 else "step_kernelILi1ELi0ELi14"
for k, c in stats.items():
    valu = sum(v for i, v in c.items() if i.startswith("v_"))
    trans = sum(v for i, v in c.items() if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", i))
    f64 = sum(v for i, v in c.items() if "_f64" in i)
    print(f"{k[:60]:60s} total {sum(c.values()):5d} valu {valu:5d} trans {trans:3d} f64 {f64:3d} salu {sum(v for i,v in c.items() if i.startswith('s_')):4d} vmem {sum(v for i,v in c.items() if i.startswith(('global_','buffer_','flat_','scratch_'))):3d} lds {sum(v for i,v in c.items() if i.startswith('ds_')):3d}")
    if want in k:
        print("   ", c.most_common(45))
