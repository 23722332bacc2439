# round 5: the judged profiles -- GPU tests, smoke, kernel trace + PMC passes of the default bench command, the driver's command shape under the trace,
# PMC of r6_step_k and of the adaptive kernel
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputest_2.log 2>&1 || { tail -30 gpurun_out/r5_gputest_2.log; exit 1; }
tail -1 gpurun_out/r5_gputest_2.log
python -c "import __graft_entry__ as g; g.smoke()"
bash scripts/profile_round.sh r5 > gpurun_out/r5_profile_round.log 2>&1 || { tail -20 gpurun_out/r5_profile_round.log; exit 1; }
tail -c 1200 gpurun_out/r5_profile_round.log
bash scripts/pmc.sh r5 --no-extras > gpurun_out/r5_pmc.log 2>&1
tail -30 gpurun_out/r5_pmc.log
bash scripts/pmc_stepk.sh r5 > gpurun_out/r5_pmc_stepk.log 2>&1
tail -5 gpurun_out/r5_pmc_stepk.log
bash scripts/pmc_rk45.sh r5 > gpurun_out/r5_pmc_rk45.log 2>&1
tail -3 gpurun_out/r5_pmc_rk45.log
