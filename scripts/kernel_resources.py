#!/usr/bin/env python3
"""Per-kernel resources of a built code object (dev tool): VGPR / AGPR / SGPR counts, spills, scratch, LDS, from the
.note metadata of the gfx950 code object inside an object file or shared library.
    python scripts/kernel_resources.py [path.o|.so] [substring]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(ROOT, "rl_rocket_6dof_amd/csrc/build/release/r6_kernels.o")
want = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    # the device code sits in the .hip_fatbin section as an offload bundle: dump it, unbundle the gfx950 code object
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={td}/fat.bin", path, f"{td}/dummy"], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={td}/fat.bin", f"--output={td}/co",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f"{td}/co"], capture_output=True, text=True).stdout
    filt = subprocess.run(["c++filt"], input=txt, capture_output=True, text=True).stdout
blocks = re.split(r"\n\s+- \.agpr_count:", filt)
for blk in blocks[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
    name = re.search(r"\.name:\s+(.*)", blk)
    name = name.group(1).strip().strip("'") if name else "?"
    if want and want not in name:
        continue
    print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>3} sgpr {g('sgpr_count'):>4} vspill {g('vgpr_spill_count'):>3} "
          f"sspill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}")
