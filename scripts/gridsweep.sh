# kbench over N x grid size (R6_GRID_BLOCKS; 0 = the library's choice), default kbench settings
for n in 1048576 2097152 4194304 8388608 16777216; do
  for g in 0 1280; do
    if [ "$g" = "0" ]; then unset R6_GRID_BLOCKS; else export R6_GRID_BLOCKS=$g; fi
    python scripts/kbench.py --n $n --configs rk4 2>&1 | tail -1 | sed "s/^/grid $g: /"
  done
done
