set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python scripts/preroll_duty_probe.py full,sync1,sync4,gap 12 > gpurun_out/r5_preroll_duty.log 2>&1
grep -v amdgpu.ids gpurun_out/r5_preroll_duty.log | sort | cut -c1-230
