#!/usr/bin/env python3
"""Forward accuracy of the step's three arithmetic routes against the fp64 oracle on the step_cfg2 fixture: composed fp32-MFMA
launches, chain kernels on the fp32 MFMA, chain kernels on the bf16 matrix cores (bf16x3).  Prints max |error| / scale of
the fused tokens, the decoder FFN pre-activation-side tensors and the three outputs.   (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load_fixture, fixture_params, fixture_batch  # noqa: E402
from tests.test_engine_gpu import build_model  # noqa: E402
from oracle import futr_oracle as O  # noqa: E402


def main():
    for tag in ("step_cfg2", "step_cfg2_zm"):
        fx = load_fixture(tag)
        m = fx["meta"]
        batch = fixture_batch(fx)
        p = {k: v.double() for k, v in fixture_params(fx).items()}
        with torch.no_grad():
            out, aux = O.forward(p, (batch[0].double(), batch[2]), batch[1].double(), "train", m["pad_idx"], m["n_head"], m["n_dec"])
        ref = dict(fused=aux["fused"].reshape(-1, m["H"]), seg=out["seg"].reshape(-1, m["n_class"]),
                   action=out["action"].reshape(-1, m["n_class"]), duration=out["duration"].reshape(-1))
        for name, flags in (("composed fp32", dict(use_fuser_chain=False, use_decoder_chain=False)),
                            ("chain fp32 MFMA", dict(chain_bf3=False)), ("chain bf16x3", dict())):
            model = build_model(fx).eval()
            eng = model.engine()
            for k, v in flags.items():
                setattr(eng, k, v)
            d = [t.cuda() for t in batch]
            o = eng.forward(d[0], d[1], d[2], "train", training=False)
            torch.cuda.synchronize()
            w = eng.last["w"]
            got = dict(fused=w.fused, seg=o["seg"].reshape(-1, m["n_class"]), action=o["action"].reshape(-1, m["n_class"]),
                       duration=o["duration"].reshape(-1))
            line = []
            for k in ("fused", "seg", "action", "duration"):
                r = ref[k]
                e = float((got[k].double().cpu() - r).abs().max()) / float(r.abs().max())
                line.append(f"{k} {e:.2e}")
            print(f"{tag:13s} {name:16s} " + "  ".join(line))


if __name__ == "__main__":
    main()
