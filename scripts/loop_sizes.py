#!/usr/bin/env python3
"""Instruction counts of a kernel of a built object / library (dev tool): total, opcode histogram, and the loops (backward
branches) with their lengths.   python scripts/loop_sizes.py path.o 'mangled-substring'"""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
path, want = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as td:
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={td}/fat.bin", path, f"{td}/dummy"], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={td}/fat.bin", f"--output={td}/co",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", f"{td}/co"], capture_output=True, text=True).stdout
    names = sorted({l.split()[-1] for l in syms.split("\n") if " FUNC " in l and want in l})
    for name in names:
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"{td}/co", f"--disassemble-symbols={name}"], capture_output=True, text=True).stdout
        ins = []
        for l in dis.split("\n"):
            m = re.match(r"\s+(\S+)\s.*//\s*([0-9A-F]+):", l)
            if m:
                ins.append((int(m.group(2), 16), m.group(1), l))
        c = collections.Counter(i[1].split("_e32")[0].split("_e64")[0] for i in ins)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print(f"{name}: {len(ins)} instructions, {valu} vector; readlane/writelane {c['v_readlane_b32']}/{c['v_writelane_b32']}, div_scale {c['v_div_scale_f32']}, "
              f"rcp {c['v_rcp_f32']}, sqrt {c['v_sqrt_f32']}, exp/log {c['v_exp_f32']}/{c['v_log_f32']}")
        addr = {a: i for i, (a, _, _) in enumerate(ins)}
        loops = []
        for i, (a, op, l) in enumerate(ins):
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", l)
                if m:
                    tgt = ins[0][0] + int(m.group(1), 16)
                    if tgt < a and tgt in addr and i - addr[tgt] > 100:
                        loops.append((addr[tgt], i))
        for a, b in loops:
            seg = ins[a:b + 1]
            cc = collections.Counter(x[1] for x in seg)
            print(f"   loop [{a}, {b}] {b - a} instr: vector {sum(v for k, v in cc.items() if k.startswith('v_'))}, readlane {cc['v_readlane_b32']}, writelane {cc['v_writelane_b32']}, s_waitcnt {cc['s_waitcnt']}")
