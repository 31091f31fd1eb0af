"""Why the N = 4H input-gradient GEMMs of the step (NN layout, K = H) take 8-10 us when the NT products of the same size
take 5-6: plain / with the gelu' or relu' + dropout epilogue, NN against NT.   python tools/nn_epi_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3d_amd import ops  # noqa: E402
from r3d_amd._lib import GEMM_NT, GEMM_NN  # noqa: E402
import bench  # noqa: E402

ws = ops.GemmWorkspace("cuda")
for M in (256, 64):
    N, K = 512, 128
    a = torch.randn(M, K, device="cuda")
    b_nn, b_nt = torch.randn(K, N, device="cuda"), torch.randn(N, K, device="cuda")
    c, aux = torch.empty(M, N, device="cuda"), torch.randn(M, N, device="cuda")
    mask = (torch.rand(M, N, device="cuda") > 0.1).to(torch.uint8)
    for name, lay, b in (("NN", GEMM_NN, b_nn), ("NT", GEMM_NT, b_nt)):
        row = []
        for label, kw in (("plain", {}), ("gelu'", dict(aux=aux, mul=2)), ("relu'+drop", dict(aux=aux, mul=1, drop_mask=mask,
                                                                                            drop_scale=1.1)),
                          ("plain t2", dict(tile=2)), ("gelu' t2", dict(aux=aux, mul=2, tile=2))):
            t = bench.time_kernel(lambda: ops.gemm(lay, a, b, c, ws=ws, **kw))
            row.append(f"{label} {t * 1e6:5.2f}")
        print(f"{name} M={M} N={N} K={K}: " + " | ".join(row), flush=True)
