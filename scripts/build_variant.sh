#!/bin/bash
# usage: scripts/build_variant.sh <name> [extra hipcc flags...]   -> rl_rocket_6dof_amd/csrc/variants/libr6dof_<name>.so
# A/B builds of the kernel library from the SAME sources with -D switches; scripts/kbench.py --libs name=path,... times
# them interleaved in one process.  (*.so is git-ignored but travels to the GPU box with gpurun.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/rl_rocket_6dof_amd/csrc/variants
mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-math-errno -fno-slp-vectorize -Wall -Wno-unused-function \
  -Wl,-Bsymbolic "$@" $ROOT/rl_rocket_6dof_amd/csrc/r6_kernels.hip $ROOT/rl_rocket_6dof_amd/csrc/r6_policy.hip $ROOT/rl_rocket_6dof_amd/csrc/r6_ppo_update.hip $ROOT/rl_rocket_6dof_amd/csrc/r6_rollout.hip -o $OUT/libr6dof_$NAME.so
echo $OUT/libr6dof_$NAME.so
