"""One-rank RCCL sanity check of the calls bench.py makes at N > 1 (init, barrier, all_reduce MAX, all_gather)."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dist.barrier()
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
rows = [torch.empty_like(t)]
dist.all_gather(rows, t)
torch.cuda.synchronize()
print("rccl ok", t.tolist(), rows[0].tolist(), dist.get_backend())
dist.barrier()
dist.destroy_process_group()
