#!/usr/bin/env python3
"""In-kernel timeline of the tiled bf16x3 weight-gradient kernel (gemm_bf3_tn_kernel, tile 10) at cfg4's per-GPU shape; needs a
build with R3D_EXTRA_DEFS=-DR3D_NT_PROBE=16 (tools/nt_timeline.sh).  Marks: consumer wave 0 at the top and the end of the MFMAs
of every k-step, producer wave 4 around every split + store + load-issue half-step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3d_amd import ops  # noqa: E402
from r3d_amd._lib import GEMM_TN  # noqa: E402

K, M, N = 512, 512, 50176
a = torch.randn(K, M, device="cuda") * 0.05
b = torch.rand(K, N, device="cuda")
cbig = torch.zeros(M + 1, N, device="cuda")               # one spare row behind C for the marks
c = cbig[:M]
ws = ops.GemmWorkspace("cuda")
rows = []
for it in range(12):
    d = ops.gemm(GEMM_TN, a, b, c, ws=ws, prec=1)
    torch.cuda.synchronize()
    marks = cbig[M, :256].view(torch.int64).cpu().double() / 100.0
    if it >= 2:
        rows.append(marks)
x = torch.stack(rows)
print(f"tile {d.tile} K {K} (k-steps {K // 32})")
for role, name in ((0, "consumer wave 0"), (1, "producer wave 4")):
    seg = x[:, role * 64: role * 64 + 64]
    base = x[:, 64]
    used = [i for i in range(60) if float(seg[:, i].max()) > 0]
    print(f"  {name}: " + "  ".join(f"[{i}] {float((seg[:, i] - base).median()):.2f}" for i in used))
