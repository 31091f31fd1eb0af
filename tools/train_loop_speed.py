#!/usr/bin/env python3
"""Steps/s of r3d_amd.train_proposed_depth.train() itself (the drop-in loop, not bench.py's captured step) on
device-resident synthetic batches of the headline shape."""
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
    val = [[t[:1] for t in make_inputs(c, dev, seed=99)]]
    args = argparse.Namespace(epochs=1, input_type="i3d_transcript", seg=True, anticipate=True, task="long",
                              graph_steps=("--graph" in sys.argv))

    class NoSched:
        def step(self): pass
    opt = FlatAdamW(model.parameters(), 1e-3, weight_decay=5e-3)
    with tempfile.TemporaryDirectory() as d:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        train(args, model, batches, opt, NoSched(), None, d, c["K"] + 1, dev, val, seed=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = len(batches)
    print(f"train(): {n} steps in {dt:.3f}s -> {dt / n * 1e6:.0f} us/step, {c['B'] * n / dt:.0f} clips/s (incl. 1 validation + checkpoint)")

if __name__ == "__main__":
    main()
