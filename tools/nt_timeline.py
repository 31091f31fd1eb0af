#!/usr/bin/env python3
"""In-kernel timeline of the forward depth projection (gemm_bf3_nt_kernel) at the headline shape; needs a build with
R3D_EXTRA_DEFS=-DR3D_NT_PROBE=16 (tools/nt_timeline.sh builds, runs and restores).  Consumer wave 0 marks the start and the
end of the MFMAs of every k-step, producer wave 4 the start and the end of every split + store + load-issue half-step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3d_amd import ops  # noqa: E402
from r3d_amd._lib import GEMM_NT  # noqa: E402

M, N, K = 128, 128, 50176
tile, splitk = int(sys.argv[1]), int(sys.argv[2])
a = torch.rand(M, K, device="cuda")
b = torch.randn(N, K, device="cuda") * 0.01
c = torch.empty(M, N, device="cuda")
ws = ops.GemmWorkspace("cuda")
ws.get(512 * M * N + 1024).zero_()
rows = []
for it in range(12):
    d = ops.gemm(GEMM_NT, a, b, c, ws=ws, prec=1, defer_reduce=True, tile=tile, splitk=splitk)
    torch.cuda.synchronize()
    marks = ws.buf[d.splitk * M * N: d.splitk * M * N + 256].view(torch.int64).cpu().double() / 100.0
    if it >= 2:
        rows.append(marks)
x = torch.stack(rows)
nk = (d.k_per_split + (32 if tile in (9, 11) else 64) - 1) // (32 if tile in (9, 11) else 64)
print(f"tile {d.tile} splitk {d.splitk} kps {d.k_per_split} k-steps {nk}")
for role, name in ((0, "consumer wave 0"), (1, "producer wave 4")):
    seg = x[:, role * 64: role * 64 + 64]
    base = x[:, 64] if tile != 11 else x[:, 0]          # mark 0 of the first-started role = kernel start of that workgroup
    used = [i for i in range(60) if float(seg[:, i].max()) > 0]
    if float(seg[:, 60].max()) > 0:                     # shader-clock counter at the loop's start and end (marks 60 / 61)
        cyc = float(((seg[:, 61] - seg[:, 60]) * 100.0).median())
        t_us = float((seg[:, 2 + 2 * nk] - seg[:, 2]).median())
        print(f"  {name}: loop {cyc:.0f} shader cycles in {t_us:.2f} us = {cyc / t_us / 1e3:.2f} GHz")
    print(f"  {name}: " + "  ".join(f"[{i}] {float((seg[:, i] - base).median()):.2f}" for i in used))
