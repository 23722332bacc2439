# maximum sizes: 2^27 + 77 envs against small shards (rows past 4 GiB)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "beyond_4_gib" > gpurun_out/r5_max_size.log 2>&1 || { tail -40 gpurun_out/r5_max_size.log; exit 1; }
tail -3 gpurun_out/r5_max_size.log
