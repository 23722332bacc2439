#!/bin/bash
# usage: scripts/build_variant.sh [--all] <name> [extra hipcc flags...]   -> rl_rocket_6dof_amd/csrc/variants/libr6dof_<name>.so
# A/B builds of the kernel library from the SAME sources with -D switches; scripts/launch_shape_ab.py / kbench.py --libs
# name=path,... time them interleaved in one process.  Only r6_kernels.hip is recompiled with the flags (the other units are the
# release objects of `python -m rl_rocket_6dof_amd.build`); --all recompiles every unit with them.
# (*.so is git-ignored but travels to the GPU box with gpurun.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ALL=0
if [ "$1" == "--all" ]; then ALL=1; shift; fi
NAME=$1; shift
C=$ROOT/rl_rocket_6dof_amd/csrc
OUT=$C/variants
OBJ=$C/build/variant_$NAME
mkdir -p $OUT $OBJ
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-math-errno -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function"   # = rl_rocket_6dof_amd/build.py FLAGS
python3 -m rl_rocket_6dof_amd.build > /dev/null
# a -D switch that is read outside r6_kernels.hip (r6_device.hpp is compiled into r6_rollout.hip too) with only r6_kernels.hip recompiled
# would link mixed objects without a word (round-4 advice): such a switch turns --all on by itself
for a in "$@"; do
  case "$a" in -D*) m=${a#-D}; m=${m%%=*};
    if [ $ALL == 0 ] && grep -q "\b$m\b" $C/r6_device.hpp $C/r6_policy_device.hpp $C/r6_rollout.hip $C/r6_policy.hip $C/r6_ppo_update.hip 2>/dev/null; then
      echo "[build_variant] $m is read outside r6_kernels.hip: recompiling every unit (--all)" >&2; ALL=1
    fi;;
  esac
done
if [ $ALL == 1 ]; then UNITS="r6_kernels r6_policy r6_ppo_update r6_rollout"; else UNITS="r6_kernels"; fi
PIDS=""
for u in $UNITS; do hipcc $FLAGS "$@" -c $C/$u.hip -o $OBJ/$u.o 2> $OBJ/$u.log & PIDS="$PIDS $!"; done
for p in $PIDS; do wait $p; done
OBJS=""
for u in r6_kernels r6_policy r6_ppo_update r6_rollout; do
  if [ -f $OBJ/$u.o ] && [[ " $UNITS " == *" $u "* ]]; then OBJS="$OBJS $OBJ/$u.o"; else OBJS="$OBJS $C/build/release/$u.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic $OBJS -o $OUT/libr6dof_$NAME.so
echo $OUT/libr6dof_$NAME.so
