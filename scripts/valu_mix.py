#!/usr/bin/env python3
"""Static VALU operand mix of the headline step kernel's code object (dev tool): how many vector instructions take a scalar-register
source (s[..], vcc, exec, m0 -- 1.8 ns per issue on gfx950 against 1.2 ns with vector / constant sources, scripts/valu_operand_bench.hip)
and how many are transcendental (3.4 ns).  Shares of the kernel's static vector instructions; bench.py applies them to the dynamic
count of the PMC pass.   python scripts/valu_mix.py [path.so|.o] [mangled-substring] -> JSON"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rl_rocket_6dof_amd/csrc/build/release/r6_kernels.o")
want = sys.argv[2] if len(sys.argv) > 2 else "step_kernelILi1ELi0ELi14ELb0ELb1"
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")
with tempfile.TemporaryDirectory() as td:
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={td}/fat.bin", path, f"{td}/dummy"], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={td}/fat.bin", f"--output={td}/co",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", f"{td}/co"], capture_output=True, text=True).stdout
    name = sorted({l.split()[-1] for l in syms.split("\n") if " FUNC " in l and want in l})[0]
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"{td}/co", f"--disassemble-symbols={name}"], capture_output=True, text=True).stdout
valu = trans = ssrc = lanes = 0
for l in dis.split("\n"):
    m = re.match(r"\s+(v_\S+)\s+(.*?)\s*//", l)
    if not m:
        continue
    op, args = m.group(1), m.group(2)
    valu += 1
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        lanes += 1
        continue
    if TRANS.match(op):
        trans += 1
        continue
    srcs = args.split(",")[1:]                      # operands after the destination
    if any(re.match(r"\s*(s\d+|s\[\d+:\d+\]|vcc|exec|m0)", a) for a in srcs) and not op.startswith("v_cmp"):
        ssrc += 1
print(json.dumps({"kernel": name, "static_vector_instructions": valu, "transcendental_share": trans / valu, "scalar_source_share": ssrc / valu,
                  "lane_move_share (v_readlane / v_writelane: SGPR spills)": lanes / valu}, indent=1))
