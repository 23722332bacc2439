set -e
timeout -k 10 300 python scripts/soak.py 100000 rk4 > gpurun_out/r4_soak_rk4.log 2>&1; tail -2 gpurun_out/r4_soak_rk4.log
timeout -k 10 300 python scripts/soak.py 20000 rk45 > gpurun_out/r4_soak_rk45.log 2>&1; tail -2 gpurun_out/r4_soak_rk45.log
timeout -k 10 300 python scripts/stepk_soak.py > gpurun_out/r4_soak_stepk.log 2>&1; tail -2 gpurun_out/r4_soak_stepk.log
