"""Forward depth-projection GEMM (NT, long K) at a BASELINE shape: planner's choice vs forced bf16x3 tiles / K-splits."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3d_amd import ops
from r3d_amd._lib import GEMM_NT
import bench
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 1024, 50176)))
a = torch.rand(M, K, device="cuda")
b = torch.randn(N, K, device="cuda") * 0.01
c = torch.empty(M, N, device="cuda")
ws = ops.GemmWorkspace("cuda")
ref = (a.double() @ b.double().t())
def run(**kw):
    d = ops.gemm(GEMM_NT, a, b, c, ws=ws, prec=1, defer_reduce=DEFER, **kw)
    return d
DEFER = False
VARIANTS = ({}, dict(tile=8, splitk=61), dict(tile=9, splitk=196), dict(tile=11, splitk=196), dict(tile=11, splitk=224), dict(tile=9, splitk=224), dict(tile=9, splitk=98),
            dict(tile=9, splitk=256), dict(tile=8, splitk=122)) if (M, N) == (128, 128) else None
for kw in VARIANTS or ({}, dict(tile=11, splitk=16), dict(tile=11, splitk=32), dict(tile=9, splitk=32), dict(tile=8, splitk=4), dict(tile=8, splitk=8), dict(tile=8, splitk=16), dict(tile=9, splitk=16), dict(tile=9, splitk=8),
           dict(tile=9, splitk=49), dict(tile=8, splitk=28)):
    try:
        d = run(**kw)
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        t = bench.time_kernel(lambda: run(**kw))
        DEFER = True
        t2 = bench.time_kernel(lambda: run(**kw))
        DEFER = False
        print(kw, "tile", d.tile, "splitk", d.splitk, "kps", d.k_per_split, "us", round(t * 1e6, 1), "without the reducer", round(t2 * 1e6, 1), "err", f"{err:.1e}", flush=True)
    except Exception as e:
        print(kw, "failed", repr(e)[:100], flush=True)
