# round 5, profiles part A: kernel trace + PMC passes (no tests: the bench's traffic / valu objects read profiles/r5_* which this job produces)
set -e
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r5 > gpurun_out/r5_profile_round.log 2>&1 || { tail -30 gpurun_out/r5_profile_round.log; exit 1; }
tail -c 1500 gpurun_out/r5_profile_round.log
bash scripts/pmc.sh r5 --no-extras > gpurun_out/r5_pmc.log 2>&1
tail -30 gpurun_out/r5_pmc.log
bash scripts/pmc_stepk.sh r5 > gpurun_out/r5_pmc_stepk.log 2>&1
tail -5 gpurun_out/r5_pmc_stepk.log
bash scripts/pmc_rk45.sh r5 > gpurun_out/r5_pmc_rk45.log 2>&1
tail -3 gpurun_out/r5_pmc_rk45.log
