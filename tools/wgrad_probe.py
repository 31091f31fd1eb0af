import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3d_amd import ops
from r3d_amd._lib import GEMM_TN
import bench
a = torch.randn(128, 128, device="cuda") * 0.05
b = torch.rand(128, 50176, device="cuda")
c = torch.empty(128, 50176, device="cuda")
ws = ops.GemmWorkspace("cuda")
for prec in (1, 0):
    t = bench.time_kernel(lambda: ops.gemm(GEMM_TN, a, b, c, ws=ws, prec=prec))
    print("prec", prec, "dbg", os.environ.get("R3D_BF3_DBG"), "us", round(t * 1e6, 2), flush=True)
