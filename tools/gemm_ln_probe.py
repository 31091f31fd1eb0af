"""Times r3d_gemm_ln_fwd against r3d_gemm_f32 + r3d_layernorm_fwd at the step's LayerNorm sites (hipGraph of 20
dependent launches, median of 5 replays) and for padded row pitches of A and W.
    python tools/gemm_ln_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3d_amd import ops  # noqa: E402


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[2]


def main():
    H = 128
    dev = "cuda"
    ws = ops.GemmWorkspace(dev)
    for M, K, pa, pw in [(256, 128, 0, 0), (64, 128, 0, 0), (256, 512, 0, 0), (256, 512, 16, 0), (256, 512, 0, 16),
                         (256, 512, 16, 16), (64, 512, 0, 0), (64, 512, 16, 16)]:
        a = torch.randn(M, K + pa, device=dev)[:, :K]
        w = (torch.randn(H, K + pw, device=dev) * K ** -0.5)[:, :K]
        bias, g, b = torch.randn(H, device=dev), torch.ones(H, device=dev), torch.zeros(H, device=dev)
        r1 = torch.randn(M, H, device=dev)
        pre, y, mean, rstd = (torch.empty(M, H, device=dev), torch.empty(M, H, device=dev), torch.empty(M, device=dev),
                              torch.empty(M, device=dev))
        job = dict(a=a, w=w, bias=bias, res1=r1, pre=pre, gamma=g, beta=b, y=y, mean=mean, rstd=rstd)
        t_new = timed(lambda: ops.gemm_ln_fwd([job]))

        def two():
            ops.gemm(0, a, w, pre, bias=bias, res1=r1, ws=ws)
            ops.layernorm_fwd(pre, g, b, y, mean, rstd)
        t_old = timed(two)
        t_g = timed(lambda: ops.gemm(0, a, w, pre, bias=bias, res1=r1, ws=ws))
        print(f"M={M} K={K} pad A/W={pa}/{pw}: gemm_ln {t_new:.2f} us | gemm + ln {t_old:.2f} us (gemm alone {t_g:.2f})",
              flush=True)


if __name__ == "__main__":
    main()
