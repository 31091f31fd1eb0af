#!/usr/bin/env python3
"""Times the pixel-sharded depth projection's two GEMMs at the per-rank shapes of W = 1, 2, 4, 8 ranks (same FLOPs at
every W): forward [W*128, P/W] x [128, P/W]^T and weight gradient [W*128, 128]^T x [W*128, P/W]; planner's choice vs a
sweep of (tile, splitk)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3d_amd import ops
from tools.gemm_sweep import time_graph

ws = ops.GemmWorkspace("cuda")
ws.get(64 * 1024 * 1024)
P, H, N = 50176, 128, 128
for W in (1, 2, 4, 8):
    for name, layout, M, Nn, K in (("fwd", 0, W * N, H, P // W), ("wgrad", 2, H, P // W, W * N)):
        sa = (M, K) if layout == 0 else (K, M)
        sb = (Nn, K) if layout == 0 else (K, Nn)
        A = torch.randn(*sa, device="cuda"); B = torch.randn(*sb, device="cuda"); C = torch.empty(M, Nn, device="cuda")
        d = ops.gemm(layout, A, B, C, ws=ws)
        auto = time_graph(lambda: ops.gemm(layout, A, B, C, ws=ws))
        best = []
        for tile in (1, 2, 3, 4, 5, 6):
            for sk in (1, 2, 4, 7, 8, 14, 16, 28, 32, 49, 64, 98, 128, 196):
                if sk > 1 and K // sk < 128:
                    continue
                try:
                    t = time_graph(lambda: ops.gemm(layout, A, B, C, ws=ws, tile=tile, splitk=sk))
                except Exception:
                    continue
                best.append((round(t, 1), tile, sk))
        best.sort()
        print(f"W={W} {name} M={M} N={Nn} K={K}: auto tile={d.tile} sk={d.splitk} {auto:.1f} us | best {best[:4]}", flush=True)

# weight gradient + AdamW on the owned columns: separate launches vs AdamW inside the GEMM's epilogue
lr_t = torch.full((1,), 1e-3, device="cuda"); step_t = torch.ones(1, dtype=torch.int64, device="cuda")
for W in (1, 2, 4, 8):
    M, Nn, K = H, P // W, W * N
    A = torch.randn(K, M, device="cuda"); B = torch.randn(K, Nn, device="cuda")
    Wt = torch.randn(M, Nn, device="cuda"); G = torch.empty(M, Nn, device="cuda")
    m = torch.zeros(M, Nn, device="cuda"); v = torch.zeros(M, Nn, device="cuda")
    adam = dict(lr_t=lr_t, step_t=step_t, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-3, grad_scale=1.0 / W, m=m, v=v)

    def separate():
        ops.gemm(2, A, B, G, ws=ws)
        ops.adamw_2d(Wt, G, m, v, lr_t, step_t, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-3, grad_scale=1.0 / W)

    def fused():
        ops.gemm(2, A, B, Wt, ws=ws, adam=adam)
    print(f"W={W} wgrad+AdamW on [{M} x {Nn}], K={K}: separate {time_graph(separate):.1f} us | fused epilogue {time_graph(fused):.1f} us",
          flush=True)
