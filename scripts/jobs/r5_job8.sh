set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 120 scripts/dep_chain_bench > gpurun_out/r5_dep_chain_bench.log 2>&1
cat gpurun_out/r5_dep_chain_bench.log
