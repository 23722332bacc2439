# kbench over N x grid size (R6_GRID_BLOCKS; 0 = the library's choice), default kbench settings
N=${1:-1048576}; shift || true
for g in ${@:-0 1024 1280 1536 2048 2560}; do
  if [ "$g" = "0" ]; then unset R6_GRID_BLOCKS; else export R6_GRID_BLOCKS=$g; fi
  python scripts/kbench.py --n $N --configs rk4 2>&1 | tail -1 | sed "s/^/grid $g: /"
done
