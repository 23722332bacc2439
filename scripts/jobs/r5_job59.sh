# the examples on the final tree: a short PPO run and the Monte-Carlo evaluation of the shipped policy
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python examples/train_ppo.py --envs 16384 --timesteps 2e8 > gpurun_out/r5_example_train.log 2>&1 || { tail -20 gpurun_out/r5_example_train.log; exit 1; }
tail -2 gpurun_out/r5_example_train.log | cut -c1-400
timeout -k 10 300 python examples/montecarlo.py --policy examples/policies/landing_policy_shared_128_64.pt --episodes 100000 > gpurun_out/r5_example_mc.log 2>&1 || { tail -20 gpurun_out/r5_example_mc.log; exit 1; }
tail -8 gpurun_out/r5_example_mc.log | cut -c1-300
