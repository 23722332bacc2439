set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest_3.log 2>&1 || { tail -30 gpurun_out/r4_gputest_3.log; exit 1; }
tail -1 gpurun_out/r4_gputest_3.log
python -c "import __graft_entry__ as g; g.smoke()"
bash scripts/profile_round.sh r4 > gpurun_out/r4_profile_round.log 2>&1 || { tail -20 gpurun_out/r4_profile_round.log; exit 1; }
tail -c 1500 gpurun_out/r4_profile_round.log
bash scripts/pmc.sh r4 --no-extras > gpurun_out/r4_pmc.log 2>&1
tail -30 gpurun_out/r4_pmc.log
bash scripts/pmc_stepk.sh r4 > gpurun_out/r4_pmc_stepk.log 2>&1
tail -5 gpurun_out/r4_pmc_stepk.log
