# host-driven stepping with the runtime polling for completion instead of sleeping on an interrupt (HSA_ENABLE_INTERRUPT=0: a process-local
# setting of the ROCm runtime): does the synchronise at the end of every single-env step get shorter?
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python scripts/single_env_rate.py > gpurun_out/r5_single_env_rate_irq.json 2> gpurun_out/r5_single_env_rate_irq.err
cat gpurun_out/r5_single_env_rate_irq.json
HSA_ENABLE_INTERRUPT=0 timeout -k 10 200 python scripts/single_env_rate.py > gpurun_out/r5_single_env_rate_poll.json 2> gpurun_out/r5_single_env_rate_poll.err
cat gpurun_out/r5_single_env_rate_poll.json
