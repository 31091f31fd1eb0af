"""The K = 4H GEMMs of the step (FFN second layers and their input gradients): planner's pick vs forced tiles."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3d_amd import ops
from r3d_amd._lib import GEMM_NT, GEMM_NN
import bench
ws = ops.GemmWorkspace("cuda")
for (lay, M, N, K) in ((GEMM_NT, 64, 128, 512), (GEMM_NT, 256, 128, 512), (GEMM_NT, 256, 512, 128), (GEMM_NT, 64, 128, 128),
                       (GEMM_NN, 64, 512, 128), (GEMM_NN, 256, 128, 512), (GEMM_NN, 64, 128, 512)):
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda") if lay == GEMM_NT else torch.randn(K, N, device="cuda")
    c = torch.empty(M, N, device="cuda")
    row = []
    for kw in ({}, dict(tile=1), dict(tile=4), dict(tile=2), dict(tile=1, splitk=2), dict(tile=1, splitk=4)):
        try:
            d = ops.gemm(lay, a, b, c, ws=ws, **kw)
            t = bench.time_kernel(lambda: ops.gemm(lay, a, b, c, ws=ws, **kw))
            row.append(f"{kw or 'plan'}: t{d.tile}/s{d.splitk} {t * 1e6:5.2f}us")
        except Exception as e:
            row.append(f"{kw}: fail")
    print(("NT" if lay == GEMM_NT else "NN"), (M, N, K), " | ".join(row), flush=True)
