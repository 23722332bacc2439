set -e
V=rl_rocket_6dof_amd/csrc/variants
python scripts/launch_shape_ab.py --libs d0=$V/libr6dof_drain0.so,d1=$V/libr6dof_drain1.so --shapes base --rounds 21 --steps 200 --alternate > gpurun_out/r4_ab_drain.log 2>&1
tail -4 gpurun_out/r4_ab_drain.log
python scripts/launch_shape_ab.py --libs d0=$V/libr6dof_drain0.so,d1=$V/libr6dof_drain1.so --shapes base --rounds 11 --steps 200 --alternate --wrapped > gpurun_out/r4_ab_drain_wrapped.log 2>&1
tail -4 gpurun_out/r4_ab_drain_wrapped.log
python scripts/launch_shape_ab.py --libs d0=$V/libr6dof_drain0.so,d1=$V/libr6dof_drain1.so --shapes base --rounds 11 --steps 200 --alternate --integrator euler > gpurun_out/r4_ab_drain_euler.log 2>&1
tail -4 gpurun_out/r4_ab_drain_euler.log
scripts/kernarg_bench > gpurun_out/r4_kernarg_bench.log 2>&1
cat gpurun_out/r4_kernarg_bench.log
