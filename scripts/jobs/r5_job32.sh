set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python scripts/soak_rk45_shapes.py > gpurun_out/r5_soak_rk45_shapes.log 2>&1 || { tail -10 gpurun_out/r5_soak_rk45_shapes.log; exit 1; }
tail -4 gpurun_out/r5_soak_rk45_shapes.log
