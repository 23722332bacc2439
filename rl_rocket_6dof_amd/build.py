"""Builds the in-tree HIP library ``csrc/libr6dof.so`` for gfx950 with hipcc (cross-compiles without
a GPU).  ``python -m rl_rocket_6dof_amd.build`` or ``__graft_entry__.build()``."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libr6dof.so")
SOURCES = ["r6_kernels.hip", "r6_policy.hip", "r6_ppo_update.hip", "r6_rollout.hip"]


def _headers():
    """Every header a source can include: csrc/*.hpp, csrc/*.h and include/*.h."""
    inc = os.path.join(HERE, "..", "include")
    return sorted(glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.h")) +
                  glob.glob(os.path.join(inc, "*.h")))


ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in [os.path.join(CSRC, s) for s in SOURCES] + _headers() + [os.path.abspath(__file__)])


FLAGS = ["-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-fno-math-errno",
         "-fno-slp-vectorize",   # packed-f32 (v_pk_*) SLP code is slower here: A/B 52 -> 46 us
         # MFMA results land in vector registers unless the register allocator itself moves an accumulator to the accumulation file.
         # Without it every matrix instruction of a kernel that needs more than 256 registers writes AGPRs, and whatever the VALU
         # consumes (tanh, the loss, the f16 splits) is copied out first: ppo_grad_kernel's tile loop 3410 -> 3188 instructions
         # (v_accvgpr_read 425 -> 169), gradient + reduce 38.1 -> 36.3 us; the rollout kernel 248 -> 171 / 304 -> 236 registers, 1-3 %.
         # The step kernels hold no matrix instruction: their code objects are byte-identical with and without.
         "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-Wall", "-Wno-unused-function"]
OBJ_DIR = os.path.join(CSRC, "build")


def _compile_objects(defines, tag: str, force: bool, verbose: bool):
    """Every source to its own object file, in parallel (one hipcc per translation unit; an unchanged unit is not rebuilt).
    Objects of a build flavour live under csrc/build/<tag>/ (git-ignored)."""
    from concurrent.futures import ThreadPoolExecutor

    odir = os.path.join(OBJ_DIR, tag)
    os.makedirs(odir, exist_ok=True)
    newest_header = max([os.path.getmtime(h) for h in _headers()] + [os.path.getmtime(os.path.abspath(__file__))])
    jobs, objs = [], []
    for src in SOURCES:
        sp, op = os.path.join(CSRC, src), os.path.join(odir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), newest_header):
            jobs.append([_hipcc(), *FLAGS, *defines, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print("[rl_rocket_6dof_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(run, jobs))
    return objs


def _link(objs, out: str, verbose: bool):
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,-Bsymbolic", *objs, "-o", out + ".tmp"]
    if verbose:
        print("[rl_rocket_6dof_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    _link(_compile_objects([], "release", force, verbose), LIB, verbose)
    _stamp_commit()
    return LIB


TEST_HOOKS_LIB = os.path.join(CSRC, "variants", "libr6dof_testhooks.so")


def build_test_hooks(force: bool = False, verbose: bool = True) -> str:
    """The same sources with -DR6_TEST_HOOKS: exports r6_internal_test_pair_abort (tests/test_gpu_step_k.py raises the device
    error word with it, as a kernel would).  The release library does not export it."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    if not force and os.path.exists(TEST_HOOKS_LIB) and all(os.path.getmtime(f) <= os.path.getmtime(TEST_HOOKS_LIB) for f in srcs):
        return TEST_HOOKS_LIB
    os.makedirs(os.path.dirname(TEST_HOOKS_LIB), exist_ok=True)
    # only the unit that owns the word differs; the other units' objects are the release ones
    objs = _compile_objects([], "release", False, verbose)
    hook_dir = os.path.join(OBJ_DIR, "testhooks")
    os.makedirs(hook_dir, exist_ok=True)
    hook_obj = os.path.join(hook_dir, "r6_kernels.o")
    cmd = [_hipcc(), *FLAGS, "-DR6_TEST_HOOKS", "-c", os.path.join(CSRC, "r6_kernels.hip"), "-o", hook_obj]
    if verbose:
        print("[rl_rocket_6dof_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    objs = [hook_obj if os.path.basename(o) == "r6_kernels.o" else o for o in objs]
    _link(objs, TEST_HOOKS_LIB, verbose)
    return TEST_HOOKS_LIB


COMMIT_FILE = os.path.join(CSRC, "BUILD_COMMIT")


def device_code_digest() -> str:
    """12 hex digits over the disassembly of the gfx950 code objects of the four release units: two libraries with the same digest run the
    same device code whatever commits their sources carry (a comment, a renamed macro).  Empty when the LLVM tools are missing."""
    import hashlib
    import tempfile

    llvm = "/opt/rocm/lib/llvm/bin"
    h = hashlib.md5()
    try:
        with tempfile.TemporaryDirectory() as td:
            for src in SOURCES:
                obj = os.path.join(OBJ_DIR, "release", src.replace(".hip", ".o"))
                subprocess.run([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={td}/fat.bin", obj, f"{td}/dummy"], check=True, capture_output=True)
                subprocess.run([f"{llvm}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={td}/fat.bin", f"--output={td}/co",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, capture_output=True)
                dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", f"{td}/co"], check=True, capture_output=True, text=True).stdout
                h.update("\n".join(ln for ln in dis.splitlines() if "file format" not in ln).encode())
        return h.hexdigest()[:12]
    except Exception:
        return ""


def _stamp_commit():
    """Records the source commit next to the library (git-ignored; travels with the .so to machines without .git, where
    bench.py reports it)."""
    try:
        root = os.path.join(HERE, "..")
        # the last commit that touched the library's sources (not HEAD: commits of docs, tests or scripts do not make a new library,
        # and profiles/ are stamped with this string)
        rev = subprocess.run(["git", "-C", root, "log", "-1", "--format=%h", "--", "rl_rocket_6dof_amd/csrc", "include", "rl_rocket_6dof_amd/build.py"], capture_output=True,
                             text=True, timeout=10).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "rl_rocket_6dof_amd/csrc", "include", "rl_rocket_6dof_amd/build.py"], capture_output=True,
                               text=True, timeout=10).stdout.strip()
        if rev:
            code = device_code_digest()
            with open(COMMIT_FILE, "w") as f:
                f.write(rev + ("+dirty" if dirty else "") + (f" code:{code}" if code else "") + "\n")
    except Exception:
        pass


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    if "--test-hooks" in sys.argv:
        print(build_test_hooks(force="--force" in sys.argv))
