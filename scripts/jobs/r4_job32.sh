set -e
timeout -k 10 800 python scripts/alloc_probe.py --envs 2 > gpurun_out/r4_alloc_probe2.log 2>&1; cut -c1-75 gpurun_out/r4_alloc_probe2.log
timeout -k 10 600 python scripts/alloc_probe.py --n 1048576 --envs 3 --steps 200 --rounds 7 --burnin 2000 --pool 2 > gpurun_out/r4_alloc_probe_1m.log 2>&1; cut -c1-75 gpurun_out/r4_alloc_probe_1m.log
