"""Depth-projection weight gradient (TN, wide N) at a BASELINE shape on the bf16x3 tiled kernel."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3d_amd import ops
from r3d_amd._lib import GEMM_TN
import bench
K, M, N = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 512, 50176)))
a = torch.randn(K, M, device="cuda") * 0.05
b = torch.rand(K, N, device="cuda")
c = torch.empty(M, N, device="cuda")
ws = ops.GemmWorkspace("cuda")
ref = a.double().t() @ b.double()
for tile in (0, 12):                         # the planner's tile (10 at these shapes) and the two-workgroups-per-CU variant
    d = ops.gemm(GEMM_TN, a, b, c, ws=ws, prec=1, tile=tile)
    err = float((c.double() - ref).abs().max() / ref.abs().max())
    t = bench.time_kernel(lambda: ops.gemm(GEMM_TN, a, b, c, ws=ws, prec=1, tile=tile))
    print("TN", (K, M, N), "tile", d.tile, "us", round(t * 1e6, 1), "err", f"{err:.1e}", flush=True)
