#!/usr/bin/env python3
"""Times r3d_gemm_f32 over (tile, splitk) for the step's dominant shapes on the GPU box (tuning aid).
Each configuration is captured into a hipGraph of 10 launches and replayed, so host launch cost does not pollute
the small shapes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3d_amd import ops

def time_graph(fn, reps=10, replays=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * replays) * 1e3   # us

def main():
    ws = ops.GemmWorkspace("cuda")
    ws.get(64 * 1024 * 1024)
    shapes = {"depth_fwd_cfg2": (0, 128, 128, 50176), "depth_wgrad_cfg2": (2, 128, 50176, 128),
              "rgb_fwd_cfg2": (0, 128, 128, 2048), "fuser_fc1_cfg2": (0, 256, 512, 128), "fuser_fc2_cfg2": (0, 256, 128, 512),
              "dec_64x128x128": (0, 64, 128, 128), "depth_fwd_cfg4": (0, 512, 512, 50176), "depth_wgrad_cfg4": (2, 512, 50176, 512)}
    shapes["sq4096"] = (0, 4096, 4096, 4096)
    shapes["sq2048_k8192"] = (0, 2048, 2048, 8192)
    for H in (128, 512, 1024):
        for N in (128, 512, 1024):
            shapes[f"dfwd_N{N}_H{H}"] = (0, N, H, 50176)
            shapes[f"dwg_N{N}_H{H}"] = (2, H, 50176, N)
            shapes[f"rgbf_N{N}_H{H}"] = (0, N, H, 2048)
            shapes[f"rgbwg_N{N}_H{H}"] = (2, H, 2048, N)
            shapes[f"fc1_N{N}_H{H}"] = (0, 2 * N, 4 * H, H)
            shapes[f"fc2_N{N}_H{H}"] = (0, 2 * N, H, 4 * H)
            shapes[f"dfc2_N{N}_H{H}"] = (1, 2 * N, 4 * H, H)
            shapes[f"dfc1_N{N}_H{H}"] = (1, 2 * N, H, 4 * H)
            shapes[f"wgfc1_N{N}_H{H}"] = (2, 4 * H, H, 2 * N)
    names = sys.argv[1:] or list(shapes)
    if names == ["grid"]:
        names = [k for k in shapes if "_N" in k]
    out = {}
    for name in names:
        layout, M, N, K = shapes[name]
        sa = (M, K) if layout in (0, 1) else (K, M)
        sb = (N, K) if layout == 0 else (K, N)
        A = torch.randn(*sa, device="cuda"); B = torch.randn(*sb, device="cuda"); Cm = torch.empty(M, N, device="cuda")
        res = []
        d = ops.gemm(layout, A, B, Cm, ws=ws)
        res.append(("auto", d.tile, d.splitk, round(time_graph(lambda: ops.gemm(layout, A, B, Cm, ws=ws)), 2)))
        for tile in (1, 2, 3, 4, 5):
            for sk in (1, 2, 4, 8, 16, 32, 49, 61, 64, 98, 122, 128, 196, 256):
                if sk > 1 and K // sk < 128: continue
                ts = {1: 32, 2: 64, 3: 128, 4: 64, 5: 128}[tile]
                if sk == 1 and K > 8192 and (M * N) // (ts * ts) < 64: continue
                try:
                    t = time_graph(lambda: ops.gemm(layout, A, B, Cm, ws=ws, tile=tile, splitk=sk, defer_reduce=True))
                except Exception as e:
                    continue
                res.append((tile, sk, round(t, 2)))
        flops = 2.0 * M * N * K
        best = sorted(res[1:], key=lambda r: r[-1])[:5]
        print(f"{name}: auto(with reduce)={res[0]} best GEMM-only (tile,splitk,us)={best} -> {flops / best[0][-1] / 1e6:.1f} TFLOP/s", flush=True)
        out[name] = res
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/gemm_sweep.json", "w"))

if __name__ == "__main__":
    main()
