#!/usr/bin/env python3
"""Times r3d_gemm_f32 over (tile, splitk) for the step's dominant shapes on the GPU box (tuning aid)."""
import sys, os, json, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3d_amd import ops

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us

def main():
    ws = ops.GemmWorkspace("cuda")
    shapes = [("depth_fwd_cfg2", 0, 128, 128, 50176), ("depth_wgrad_cfg2", 2, 128, 50176, 128),
              ("rgb_fwd_cfg2", 0, 128, 128, 2048), ("rgb_wgrad_cfg2", 2, 128, 2048, 128),
              ("fuser_fc1_cfg2", 0, 256, 512, 128), ("fuser_fc2_cfg2", 0, 256, 128, 512),
              ("depth_fwd_cfg4", 0, 512, 512, 50176), ("depth_wgrad_cfg4", 2, 512, 50176, 512)]
    if len(sys.argv) > 1:
        shapes = [s for s in shapes if s[0] in sys.argv[1:]]
    out = {}
    for name, layout, M, N, K in shapes:
        sa = (M, K) if layout in (0, 1) else (K, M)
        sb = (N, K) if layout == 0 else (K, N)
        A = torch.randn(*sa, device="cuda"); B = torch.randn(*sb, device="cuda"); Cm = torch.empty(M, N, device="cuda")
        res = []
        d = ops.gemm(layout, A, B, Cm, ws=ws)
        t = timeit(lambda: ops.gemm(layout, A, B, Cm, ws=ws))
        res.append(("auto", d.tile, d.splitk, round(t, 2)))
        for tile in (1, 2, 3):
            for sk in (1, 2, 4, 8, 16, 32, 49, 64, 98, 128):
                if sk > 1 and K // sk < 64: continue
                if sk == 1 and K > 8192 and (M * N) // (32 * 32 * (4 ** (tile - 1))) < 64: continue
                try:
                    t = timeit(lambda: ops.gemm(layout, A, B, Cm, ws=ws, tile=tile, splitk=sk), iters=10, warm=2)
                except Exception as e:
                    continue
                res.append((tile, sk, round(t, 2)))
        flops = 2.0 * M * N * K
        best = min(res[1:], key=lambda r: r[-1])
        print(f"{name}: auto={res[0]} best(tile,splitk,us)={best} -> {flops / best[-1] / 1e6:.1f} TFLOP/s", flush=True)
        out[name] = res
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/gemm_sweep.json", "w"))

if __name__ == "__main__":
    main()
