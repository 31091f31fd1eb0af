#!/usr/bin/env python3
"""Launches only the two depth-projection GEMMs of the bench shape (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3d_amd import ops
N, H, P = 128, 128, 50176
X = torch.rand(N, P, device="cuda"); W = torch.randn(H, P, device="cuda") * 0.01
dY = torch.randn(N, H, device="cuda"); out = torch.empty(N, H, device="cuda"); dW = torch.empty(H, P, device="cuda")
ws = ops.GemmWorkspace("cuda")
for _ in range(5):
    ops.gemm(0, X, W, out, ws=ws, defer_reduce=True, prec=1)      # the kernels the engine runs (depth_prec = 1)
    ops.gemm(2, dY, X, dW, ws=ws, prec=1)
torch.cuda.synchronize()
print("done")
