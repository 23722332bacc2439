set -e
cd $GRAFT_REPO_ROOT
bash scripts/jobs/r5_profiles_a.sh
