set -e
V=rl_rocket_6dof_amd/csrc/variants
L="old=$V/libr6dof_drain0.so,hot0=$V/libr6dof_hot0.so,pre=$V/libr6dof_hot1.so,nopre=$V/libr6dof_hot1nopre.so,rot=$V/libr6dof_rot.so,inl2=$V/libr6dof_inloop2.so"
python scripts/launch_shape_ab.py --libs $L --shapes base --rounds 21 --steps 200 --alternate > gpurun_out/r4_ab_hot.log 2>&1
grep -v raw gpurun_out/r4_ab_hot.log | tail -8
python scripts/launch_shape_ab.py --libs $L --shapes base --rounds 11 --steps 200 --alternate --integrator euler --nocheck > gpurun_out/r4_ab_hot_euler.log 2>&1
grep -v raw gpurun_out/r4_ab_hot_euler.log | tail -6
python scripts/launch_shape_ab.py --libs $L --shapes base --rounds 11 --steps 200 --alternate --wrapped > gpurun_out/r4_ab_hot_wrapped.log 2>&1
grep -v raw gpurun_out/r4_ab_hot_wrapped.log | tail -6
