# the ranks' own stdout under torch.distributed.run (the way the DRIVER starts an N > 1 job: no launcher of ours in between): one JSON line, no gloo banner
set -e
cd $GRAFT_REPO_ROOT
export R6_DIST_BACKEND=gloo R6_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 2 --envs-per-gpu 262144 --steps 20 --warmup 5 > gpurun_out/r5_torchrun_2rank.out 2> gpurun_out/r5_torchrun_2rank.err
echo "stdout lines: $(wc -l < gpurun_out/r5_torchrun_2rank.out)"; cut -c1-200 gpurun_out/r5_torchrun_2rank.out
grep -c "Gloo" gpurun_out/r5_torchrun_2rank.err || true
