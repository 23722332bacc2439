set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python scripts/preroll_probe.py 0.25,0.5,1.0,1.5,2.5,4.0 8 2.0 > gpurun_out/r5_preroll_probe.log 2>&1
grep -v amdgpu.ids gpurun_out/r5_preroll_probe.log | tail -60
