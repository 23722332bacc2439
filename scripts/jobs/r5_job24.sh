set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/debug_guests.py > gpurun_out/r5_debug_guests.log 2>&1 || true
tail -40 gpurun_out/r5_debug_guests.log
