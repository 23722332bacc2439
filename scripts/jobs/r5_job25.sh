set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/debug_guests.py > gpurun_out/r5_debug_guests.log 2>&1 || true
grep -c "differ" gpurun_out/r5_debug_guests.log || true
grep "differ" gpurun_out/r5_debug_guests.log | head -5 | cut -c1-200
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "straggler" > gpurun_out/r5_rk45_guest_test.log 2>&1 || { tail -30 gpurun_out/r5_rk45_guest_test.log; exit 1; }
tail -2 gpurun_out/r5_rk45_guest_test.log
