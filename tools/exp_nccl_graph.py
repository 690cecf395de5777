"""Experiment: can an RCCL all-reduce (torch.distributed, backend nccl) be captured in a hipGraph and replayed?
Run on the 1-GPU box with a 1-rank process group."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)                      # eager warm-up creates the communicator
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        y = x * 2
        w = dist.all_reduce(y, async_op=True)
        w.wait()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        y = x * 2
        w = dist.all_reduce(y, async_op=True)
        z = x + 1                       # independent work that may overlap
        w.wait()
        out = y + z
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("NCCL-in-graph OK", float(out[0]))
except Exception as e:                   # noqa: BLE001
    print("NCCL-in-graph FAILED:", repr(e)[:500])
dist.destroy_process_group()
