"""The abort path of step_k_pair_kernel, provoked (dev tool, GPU box): a build whose bounded waits give up after 8 polls
    scripts/build_variant.sh abort -DR6_PAIR_SPIN_LOG2=3 && python scripts/stepk_abort_check.py
must make the NEXT r6_step_k call fail with R6_E_DEVICE; the default build must run the same calls clean."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config, _lib

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name, path, want_abort in (("default", None, False), ("abort build", os.path.join(root, "rl_rocket_6dof_amd/csrc/variants/libr6dof_abort.so"), True)):
    env = BatchedRocket6DOF(65536, default_env_config(), integrator="euler", timestep=0.04, seed=3, lib_path=path)
    env.reset()
    act = torch.rand((16, 65536, 3), device=env.device) * 2 - 1
    err = None
    try:
        for _ in range(50):
            env.step_k(act)
            torch.cuda.synchronize()
    except _lib.R6Error as e:
        err = str(e)
    print(f"{name}: {'R6Error: ' + err[:160] if err else 'no error in 50 launches of 16 steps'}")
    assert (err is not None and "-10003" in err) == want_abort, (name, err)
print("ok")
