set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputest_final.log 2>&1 || { tail -40 gpurun_out/r5_gputest_final.log; exit 1; }
tail -1 gpurun_out/r5_gputest_final.log
grep "driver shape\]" gpurun_out/r5_gputest_final.log || true
python -c "import __graft_entry__ as g; g.smoke()"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_shape_final.json 2> gpurun_out/r5_bench_driver_shape_final.err
python3 -c "
import json; b=json.load(open('gpurun_out/r5_bench_driver_shape_final.json')); print('driver shape: wall', round(b['ms_per_step']*1e3,2), 'kernel', round(b['roofline']['kernel_us'],2), 'frac', round(b['roofline']['frac'],3), 'steady median', round(b['steady']['median'],2), 'lib', b['config']['lib_commit'])"
