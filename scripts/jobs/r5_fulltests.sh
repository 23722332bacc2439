set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputest_1.log 2>&1 || { tail -60 gpurun_out/r5_gputest_1.log; exit 1; }
tail -3 gpurun_out/r5_gputest_1.log
python -c "import __graft_entry__ as g; g.smoke()"
