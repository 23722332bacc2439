for cfg in "2097152 1280" "2097152 2560" "4194304 1280" "4194304 2560" "4194304 5120" "8388608 1280" "8388608 5120" "8388608 10240" "16777216 16384"; do
  set -- $cfg
  R6_GRID_BLOCKS=$2 python scripts/kbench.py --n $1 --configs rk4 --steps 40 --rounds 3 2>&1 | tail -1 | sed "s/^/grid $2: /"
done
