import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
n = 1 << 20
env = BatchedRocket6DOF(n, default_env_config(), record_flags=False, track_episodes=True, seed=42)
gen = torch.Generator(device=env.device).manual_seed(1)
pool = [(torch.rand((n, 3), generator=gen, device=env.device) * 2 - 1) for _ in range(4)]
env.reset()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(601)]
dones = []
evs[0].record()
for k in range(600):
    env.step_async(pool[k % 4])
    evs[k + 1].record()
    if k % 25 == 24:
        dones.append(int(env.done.sum().item()))
torch.cuda.synchronize()
ts = [evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(600)]
for b in range(0, 600, 25):
    seg = ts[b:b + 25]
    print(f"steps {b:3d}-{b+24:3d}: mean {sum(seg)/len(seg):6.1f} us  min {min(seg):6.1f}  done/step at end {dones[b//25]}")
