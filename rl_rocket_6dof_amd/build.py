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
    return any(os.path.getmtime(f) > t for f in [os.path.join(CSRC, s) for s in SOURCES] + _headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), "-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-shared",
           "-fno-math-errno", "-fno-slp-vectorize",   # packed-f32 (v_pk_*) SLP code is slower here: A/B 52 -> 46 us
           "-Wall", "-Wno-unused-function", "-Wl,-Bsymbolic",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp"]
    if verbose:
        print("[rl_rocket_6dof_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    _stamp_commit()
    return LIB


COMMIT_FILE = os.path.join(CSRC, "BUILD_COMMIT")


def _stamp_commit():
    """Records the source commit next to the library (git-ignored; travels with the .so to machines without .git, where
    bench.py reports it)."""
    try:
        root = os.path.join(HERE, "..")
        rev = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "rl_rocket_6dof_amd/csrc", "include"], capture_output=True,
                               text=True, timeout=10).stdout.strip()
        if rev:
            with open(COMMIT_FILE, "w") as f:
                f.write(rev + ("+dirty" if dirty else "") + "\n")
    except Exception:
        pass


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
