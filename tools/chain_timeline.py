#!/usr/bin/env python3
"""In-kernel stage timelines of the chain kernels (fuser_chain.hip, decoder_chain.hip) at the bench shape: wave 0 of the
first workgroup of each role stores wall_clock64() (100 MHz) at its stage boundaries; prints the deltas in microseconds,
median over steps.   python tools/chain_timeline.py [--graph]   (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    c = dict(bench.CFG)
    model = bench.build_model(c, dev)
    eng = model.engine()
    feats, depth, lab, dur, tgt = bench.make_inputs(c, dev, 1)
    for _ in range(3):
        eng.train_step(feats, depth, lab, dur, tgt, c["lr"], c["wd"], training=True)
    torch.cuda.synchronize()
    w = eng.last["w"]
    tl = {}
    for key, obj in w.tables.items():
        if key[0] in ("fwd_chain", "bwd_chain", "dec_chain") and hasattr(obj, "args"):
            if key[0] in tl:
                continue
            t = torch.zeros(32, dtype=torch.int64, device=dev)
            obj.args.timeline = t.data_ptr()
            tl[key[0]] = t
    graph = "--graph" in sys.argv
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            eng.train_step(feats, depth, lab, dur, tgt, c["lr"], c["wd"], training=True)
    rec = {k: [] for k in tl}
    for _ in range(20):
        if graph:
            g.replay()
        else:
            eng.train_step(feats, depth, lab, dur, tgt, c["lr"], c["wd"], training=True)
        torch.cuda.synchronize()
        for k, t in tl.items():
            rec[k].append(t.cpu().clone())
    for k, rows in rec.items():
        x = torch.stack(rows).double() / 100.0            # us
        print(f"== {k} ({'graph replay' if graph else 'eager'})")
        for lo, hi, name in ((0, 16, "role 0"), (16, 32, "role 1")):
            seg = x[:, lo:hi]
            used = [i for i in range(hi - lo) if float(seg[:, i].max()) > 0]
            if len(used) < 2:
                continue
            base = seg[:, used[0]]
            rel = {i: float((seg[:, i] - base).median()) for i in used}
            order = sorted(used, key=lambda i: rel[i])
            line = [f"[{lo + i}] {rel[i]:.2f}" for i in order]
            print(f"   {name} (us since the first mark): " + "  ".join(line))


if __name__ == "__main__":
    main()
