# the driver's command shape with the pre-roll in batches of 4 launches (92 % duty): eight fresh processes, then the old form (batches of 200) four times
set -e
cd $GRAFT_REPO_ROOT
: > gpurun_out/r5_driver_shape_b4.log
for i in 1 2 3 4 5 6 7 8; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r5_ds_b4_$i.json 2> gpurun_out/r5_ds_b4_$i.err
  python3 -c "
import json; b=json.load(open('gpurun_out/r5_ds_b4_$i.json')); s=b['steady']; print('batch 4   run $i: wall', round(b['ms_per_step']*1e3,2), 'kernel', round(b['roofline']['kernel_us'],2), 'frac', round(b['roofline']['frac'],3), 'steady median', round(s['median'],2), 'p10', round(s['p10'],2), 'p90', round(s['p90'],2), 'ratio', round(s['timed_region_over_median'],3))" >> gpurun_out/r5_driver_shape_b4.log
done
for i in 1 2 3 4; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --preroll-batch 200 > gpurun_out/r5_ds_b200_$i.json 2> gpurun_out/r5_ds_b200_$i.err
  python3 -c "
import json; b=json.load(open('gpurun_out/r5_ds_b200_$i.json')); s=b['steady']; print('batch 200 run $i: wall', round(b['ms_per_step']*1e3,2), 'kernel', round(b['roofline']['kernel_us'],2), 'frac', round(b['roofline']['frac'],3), 'steady median', round(s['median'],2), 'p10', round(s['p10'],2), 'p90', round(s['p90'],2), 'ratio', round(s['timed_region_over_median'],3))" >> gpurun_out/r5_driver_shape_b4.log
done
python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/r5_default_b4.json 2> gpurun_out/r5_default_b4.err
python3 -c "
import json; b=json.load(open('gpurun_out/r5_default_b4.json')); s=b['steady']; print('default (2000 steps): wall', round(b['ms_per_step']*1e3,2), 'kernel', round(b['roofline']['kernel_us'],2), 'frac', round(b['roofline']['frac'],3), 'steady median', round(s['median'],2), 'p10', round(s['p10'],2), 'p90', round(s['p90'],2))" >> gpurun_out/r5_driver_shape_b4.log
cat gpurun_out/r5_driver_shape_b4.log
