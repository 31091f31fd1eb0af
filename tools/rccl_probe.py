"""One-rank probe of r3d_amd/rccl.py: every collective eagerly, then captured into a hipGraph and replayed."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
from r3d_amd.rccl import RcclComm

c = RcclComm()
print("comm ok", c.world, c.rank, flush=True)
x = torch.arange(1024, dtype=torch.float32, device=dev)
c.all_reduce(x); torch.cuda.synchronize(); print("all_reduce", float(x.sum()), flush=True)
y = torch.arange(1024, dtype=torch.float32, device=dev)
m = c.reduce_scatter_inplace(y); torch.cuda.synchronize(); print("reduce_scatter", float(m.sum()), flush=True)
o = torch.zeros(1024, dtype=torch.float32, device=dev)
c.all_gather(o, x); torch.cuda.synchronize(); print("all_gather", float(o.sum()), flush=True)
r = torch.zeros(1024, dtype=torch.float32, device=dev)
c.all_to_all(r, x); torch.cuda.synchronize(); print("all_to_all", float(r.sum()), flush=True)
with c.group():
    c.all_gather(o, x)
    c.all_reduce(x)
torch.cuda.synchronize(); print("group ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    with c.group():
        c.all_gather(o, x)
        c.all_reduce(y)
    c.all_reduce(x)
    c.reduce_scatter_inplace(y)
    c.all_gather(o, x)
    c.all_to_all(r, x)
g.replay(); torch.cuda.synchronize(); print("graph ok", float(r.sum()), flush=True)
dist.destroy_process_group()
