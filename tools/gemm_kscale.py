import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3d_amd import ops
from tools.gemm_sweep import time_graph
ws = ops.GemmWorkspace("cuda"); ws.get(64*1024*1024)
M=N=128
for K in (6272, 12544, 25088, 50176, 100352):
    A=torch.randn(M,K,device="cuda"); B=torch.randn(N,K,device="cuda"); C=torch.empty(M,N,device="cuda")
    for tile, sk in ((2,61),(2,16),(2,8),(4,61)):
        t=time_graph(lambda: ops.gemm(0,A,B,C,ws=ws,tile=tile,splitk=sk,defer_reduce=True))
        steps = K/sk/64
        print(f"K={K:6d} tile={tile} splitk={sk:3d} steps/WG={steps:6.1f}  {t:7.2f} us  {2*M*N*K/t/1e6:6.1f} TF", flush=True)
