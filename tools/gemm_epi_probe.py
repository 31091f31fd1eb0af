#!/usr/bin/env python3
"""How much of a tiny GEMM launch is its epilogue?  hipGraph-timed chains of the step's small shapes with growing
epilogues (tuning aid, GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3d_amd import ops
from tools.gemm_sweep import time_graph

def main():
    ws = ops.GemmWorkspace("cuda")
    M, N, K = 256, 128, 128
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda"); r1 = torch.randn(M, N, device="cuda"); r2 = torch.randn(M, N, device="cuda")
    aux = torch.randn(M, N, device="cuda"); pre = torch.empty(M, N, device="cuda")
    dm = (torch.rand(M, N, device="cuda") > 0.1).to(torch.uint8)
    Bn = torch.randn(K, N, device="cuda")
    cases = {
        "NT plain": lambda: ops.gemm(0, A, B, C, ws=ws),
        "NT bias": lambda: ops.gemm(0, A, B, C, bias=bias, ws=ws),
        "NT bias+res1": lambda: ops.gemm(0, A, B, C, bias=bias, res1=r1, ws=ws),
        "NT bias+res1+res2": lambda: ops.gemm(0, A, B, C, bias=bias, res1=r1, res2=r2, ws=ws),
        "NT bias+gelu+pre": lambda: ops.gemm(0, A, B, C, bias=bias, act=2, pre_out=pre, ws=ws),
        "NT bias+drop+res1": lambda: ops.gemm(0, A, B, C, bias=bias, drop_mask=dm, drop_scale=1.1, res1=r1, ws=ws),
        "NN plain": lambda: ops.gemm(1, A, Bn, C, ws=ws),
        "NN aux mul": lambda: ops.gemm(1, A, Bn, C, aux=aux, mul=2, ws=ws),
        "NN res1": lambda: ops.gemm(1, A, Bn, C, res1=r1, ws=ws),
    }
    for k, fn in cases.items():
        print(f"{k:22s} {time_graph(fn, reps=20):7.2f} us", flush=True)
    x = torch.randn(M, N, device="cuda"); y = torch.empty_like(x); g = torch.ones(N, device="cuda"); b = torch.zeros(N, device="cuda")
    mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    print(f"{'ln_fwd 256x128':22s} {time_graph(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd), reps=20):7.2f} us")
    t = torch.zeros(1, dtype=torch.int64, device="cuda")
    print(f"{'tick (empty kernel)':22s} {time_graph(lambda: ops.tick(t, None), reps=20):7.2f} us")

if __name__ == "__main__":
    main()
