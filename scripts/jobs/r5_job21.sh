set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 120 scripts/grid_barrier_bench > gpurun_out/r5_grid_barrier_bench.log 2>&1
cat gpurun_out/r5_grid_barrier_bench.log
