# timing-only experiment: the variants were built from a working copy with the Newton loop of t_go_quartic (r6_device.hpp) cut to 6 / 4
# iterations behind a throw-away -DR6_TGO_ITERS switch, and with -DR6_ABLATE_REWARD; the switch was not kept
set -e
cd $GRAFT_REPO_ROOT
V=rl_rocket_6dof_amd/csrc/variants
timeout -k 10 600 python scripts/ab_swap.py --libs base=rl_rocket_6dof_amd/csrc/libr6dof.so,tgo6=$V/libr6dof_tgo6.so,tgo4=$V/libr6dof_tgo4.so,norew=$V/libr6dof_norew.so --envs 2 --rounds 15 --steps 200 --reset-on-swap 300 > gpurun_out/r5_swap_tgo.log 2>&1
grep POOLED gpurun_out/r5_swap_tgo.log | cut -c1-160
