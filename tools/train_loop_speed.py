#!/usr/bin/env python3
"""Steps/s of r3d_amd.train_proposed_depth.train() itself (the drop-in loop, not bench.py's captured step) on synthetic
batches of the headline shape: device-resident (default) or `--host`: pinned HOST batches through
r3d_amd.utils.InputPrefetcher -- the PCIe-inclusive rate (27 MB of inputs per step cross the bus under the previous step).
`--npy`: the same batches as per-clip `.npy` files (in /dev/shm: page-cache resident) read by r3d_amd.utils.NpyClipReader into
pinned staging buffers, then InputPrefetcher -- the whole real-data input path minus the disk.
    python tools/train_loop_speed.py [--graph] [--host | --npy]"""
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
    npy = "--npy" in sys.argv
    shm = None
    if npy:
        import numpy as np, shutil
        from r3d_amd.utils import InputPrefetcher, NpyClipReader
        shm = tempfile.mkdtemp(dir="/dev/shm")
        host4 = [[t.cpu() for t in b] for b in batches[:4]]
        specs = []
        for i, b in enumerate(host4):
            clips = []
            for k in range(b[0].shape[0]):
                fp, dp = os.path.join(shm, f"b{i}c{k}.npy"), os.path.join(shm, f"b{i}c{k}_1.npy")
                np.save(fp, b[0][k].numpy()); np.save(dp, b[1][k].numpy())
                clips.append((fp, dp, 0, b[0].shape[1], 1))
            specs.append(clips)
        rd = NpyClipReader()

        class Loader:
            def __iter__(self):
                for j in range(2000):
                    f, d = rd.batch(specs[j % 4])
                    yield [f, d] + [t.pin_memory() if not t.is_pinned() else t for t in host4[j % 4][2:]]

            def __len__(self):
                return 2000
        batches = InputPrefetcher(Loader(), dev)
        host = True
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
    if shm:
        shutil.rmtree(shm, ignore_errors=True)
    print((".npy files -> NpyClipReader (pinned) -> InputPrefetcher, " if npy else "host batches over PCIe (InputPrefetcher), " if host else "device-resident batches, ") +
          ("hipGraph steps: " if "--graph" in sys.argv else "enqueued steps: "), end="")
    print(f"train(): {n} steps in {dt:.3f}s -> {dt / n * 1e6:.0f} us/step, {c['B'] * n / dt:.0f} clips/s (incl. 1 validation + checkpoint)")

if __name__ == "__main__":
    main()
