#!/usr/bin/env python3
"""Round-2 measurement harness (runs on the GPU box, alone or under rocprofv3):

    python tools/r02_profile.py erank [--out gpurun_out/x/erank.json]     effective-rank Jacobi at the BASELINE shapes
    python tools/r02_profile.py adamw [--out ...]                          AdamW at the cfg2 / cfg4 / cfg5 arena sizes

erank: the fused token matrix [N, H] of BASELINE.json configs[1], [3] and [4] (cfg2 [128,128]; cfg4 per-GPU [512,512];
cfg5 [2048,1024] global and [256,1024] per-GPU).  The reference has no SVD: the checker is
torch.linalg.svdvals on the CPU copy of the same matrix.  Per shape: route, sweeps, HIP-event time (median of 5),
us per sweep, and the column-sweep traffic the kernel's loop structure implies (for the LDS-resident
kernel the figure is LDS bytes; for the blocked kernel every cross round reads and writes each column once, i.e.
2 * R * C * 4 bytes per round through L2 / HBM).
adamw: 28 B/param over the live arena at 7.4 M / 43.8 M / 119.7 M parameters -- the last two
exceed the 256 MiB Infinity Cache, so their GB/s is an HBM figure.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [("cfg2 fused tokens", 128, 128), ("cfg4 per-GPU", 512, 512), ("cfg5 per-GPU (B=16)", 256, 1024),
          ("cfg5 global (B=128)", 2048, 1024)]


def token_like(R, C, seed):
    """A matrix with the spectrum shape of fused tokens at init: one dominant direction (the LayerNorm/ReLU mean), a
    decaying bulk, full numerical rank min(R, C)."""
    g = torch.Generator().manual_seed(seed)
    k = min(R, C)
    u = torch.linalg.qr(torch.randn(R, k, generator=g))[0]
    v = torch.linalg.qr(torch.randn(C, k, generator=g))[0]
    s = torch.exp(-torch.arange(k, dtype=torch.float32) / (0.35 * k)) + 0.02
    s[0] = 6.0
    return (u * s) @ v.t()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]          # us


def timed_graph(fn, reps=5):
    """The same launches replayed as a hipGraph (how they run inside the training step: no host dispatch between the
    dependent launches of the two-level kernel).  None if the capture fails."""
    try:
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return timed(g.replay, reps)
    except Exception:                                       # noqa: BLE001
        torch.cuda.synchronize()
        return None


def run_erank(a):
    from r3d_amd import ops, erank as ER
    dev = torch.device("cuda:0")
    rows = []
    for name, R, C in SHAPES:
        x_cpu = token_like(R, C, 7)
        x = x_cpu.to(dev)
        sv = torch.linalg.svdvals(x_cpu.double())          # LAPACK on the host: the checker, outside any timing
        pr = sv / sv.sum()
        pr = pr[pr > 0]
        want = float(torch.exp(-(pr * pr.log()).sum()))
        for route in a.routes.split(","):
            if route == "lds" and not ops.erank_fits(R, C):
                continue
            if route == "blocked":
                xx = x.t().contiguous() if R < C else x
                state = {}

                def fn():
                    state["r"] = ops.erank_blocked(xx)
                us = timed(fn)
                sigma, stats, _ = state["r"]
                st = stats.cpu()
                Rr, Cc = xx.shape
                info = ops.erank_blocked_info(Rr, Cc) if hasattr(ops, "erank_blocked_info") else {}
                info = dict(info, us_in_graph=timed_graph(fn))
            elif route == "lds":
                sigma = torch.empty(1, C, device=dev)
                stats = torch.empty(1, 4, device=dev)
                af = torch.empty(1, C, R, device=dev)
                us = timed(lambda: ops.erank_jacobi(x, sigma, stats, af_t=af))
                st = stats[0].cpu()
                Rr, Cc = R, C
                info = {}
            else:
                fnr = getattr(ER, "effective_rank_" + route, None)
                if fnr is None:
                    continue
                state = {}

                def fn():
                    state["r"] = fnr(x)
                us = timed(fn)
                st = state["r"].cpu()
                Rr, Cc = (C, R) if R < C else (R, C)
                info = {}
            sweeps = float(st[3])
            got = float(st[0])
            # one sweep rotates every column pair once: each round touches all Cc columns (read + write)
            rounds = Cc - 1 + (Cc & 1)
            sweep_bytes = 2.0 * Rr * Cc * 4 * rounds
            row = dict(shape=name, R=R, C=C, route=route, erank_hip=got, erank_svdvals=want, abs_diff=abs(got - want),
                       sweeps=sweeps, us=us, us_per_sweep=us / max(sweeps, 1.0),
                       column_sweep_GBps=sweep_bytes * sweeps / us / 1e3, **info)
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


def run_adamw(a):
    from r3d_amd import ops
    import bench
    dev = torch.device("cuda:0")
    lr_t = torch.full((1,), 1e-3, device=dev)
    step_t = torch.ones(1, dtype=torch.int64, device=dev)
    rows = []
    for name, n in (("cfg2 (H=128)", 7_410_472), ("cfg4 (H=512)", 35_154_000), ("cfg5 (H=1024)", 85_000_000)):
        n = n // 4 * 4
        p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
        v.abs_()
        t = bench.time_kernel(lambda: ops.adamw_flat(p, g, m, v, lr_t, step_t, weight_decay=5e-3))
        row = dict(arena=name, live_params=n, bytes=28.0 * n, working_set_MiB=16.0 * n / 2**20, us=t * 1e6,
                   GBps=28.0 * n / t / 1e9, frac_of_8TBps=28.0 * n / t / 8e12)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del p, g, m, v
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["erank", "adamw"])
    ap.add_argument("--routes", default="lds,blocked")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    rows = run_erank(a) if a.what == "erank" else run_adamw(a)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
