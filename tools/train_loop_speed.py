#!/usr/bin/env python3
"""Steps/s of r3d_amd.train_proposed_depth.train() itself (the drop-in loop, not bench.py's captured step) on synthetic
batches of the headline shape: device-resident (default) or `--host`: pinned HOST batches through
r3d_amd.utils.InputPrefetcher -- the PCIe-inclusive rate (27 MB of inputs per step cross the bus under the previous step).
    python tools/train_loop_speed.py [--graph] [--host]"""
import argparse, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import CFG, make_inputs, build_model
from r3d_amd.train_proposed_depth import train
from r3d_amd.optim import FlatAdamW

def main():
    dev = torch.device("cuda", 0)
    c = CFG
    model = build_model(c, dev)
    batches = [make_inputs(c, dev, seed=s) for s in range(4)] * 500         # 2000 steps per epoch
    host = "--host" in sys.argv
    if host:
        from r3d_amd.utils import InputPrefetcher
        pinned = [[t.cpu().pin_memory() for t in b] for b in batches[:4]]
        batches = InputPrefetcher(pinned * 500, dev)
    val = [[t[:1] for t in make_inputs(c, dev, seed=99)]]
    args = argparse.Namespace(epochs=1, input_type="i3d_transcript", seg=True, anticipate=True, task="long",
                              graph_steps=("--graph" in sys.argv))

    class NoSched:
        def step(self): pass
    opt = FlatAdamW(model.parameters(), 1e-3, weight_decay=5e-3)
    with tempfile.TemporaryDirectory() as d:
        # first call: allocations, GEMM planning, graph capture; the second call is the steady state that is reported
        train(args, model, batches, opt, NoSched(), None, d, c["K"] + 1, dev, val, seed=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        train(args, model, batches, opt, NoSched(), None, d, c["K"] + 1, dev, val, seed=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = len(batches)
    print(("host batches over PCIe (InputPrefetcher), " if host else "device-resident batches, ") +
          ("hipGraph steps: " if "--graph" in sys.argv else "enqueued steps: "), end="")
    print(f"train(): {n} steps in {dt:.3f}s -> {dt / n * 1e6:.0f} us/step, {c['B'] * n / dt:.0f} clips/s (incl. 1 validation + checkpoint)")

if __name__ == "__main__":
    main()
