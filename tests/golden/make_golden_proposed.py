#!/usr/bin/env python3
"""Generates tests/golden/proposed_*.npz by IMPORTING THE REFERENCE's label-query model (model/futr_proposed.py) on CPU --
build container only, no shim needed.  Its loop (train/train_unsupervised.py) is out of scope, so the
fixture pins the module itself: forward in eval() state with mode='train' on hash-generated inputs / analytic parameters,
and the gradients of a fixed differentiable function of the three outputs through the reference's own autograd.  Every
value is cross-checked against oracle/futr_oracle.py: forward_proposed."""
import importlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path[:0] = ["/root/reference", "/root/reference/train"]

from oracle import synth, futr_oracle as O  # noqa: E402
from opts import parser  # noqa: E402
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

M = importlib.import_module("model.futr_proposed")


def probe_loss(out):
    return (out["seg"] ** 2).mean() + 0.01 * out["action"].sum() + (0.1 * out["duration"]).exp().mean()


def case(tag, H, B, S, n_class, n_dec, query_num, seed):
    args = parser.parse_args([])
    args.hidden_dim, args.n_head, args.n_decoder_layer, args.n_query = H, 8, n_dec, 8
    pad_idx = n_class + 1
    model = M.FUTR(n_class, H, device=torch.device("cpu"), args=args, src_pad_idx=pad_idx, n_query=8, n_head=8,
                   num_encoder_layers=args.n_encoder_layer, num_decoder_layers=n_dec, query_num=query_num)
    names_shapes = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    state = synth.fill_state(names_shapes)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[n]))
    model.eval()
    feats, _, lab, _, _ = G.t_batch(synth.make_batch(B, S, n_class, pad_idx, seed, depth_hw=(2, 2)))
    query = torch.from_numpy(synth.randint(B * S, query_num, (seed << 8) + 77).reshape(B, S))
    out = model((feats, lab), query)
    loss = probe_loss(out)
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters()}
    live = [n for n, g in grads.items() if g is not None]
    fx = {
        "meta": json.dumps(dict(tag=tag, H=H, B=B, S=S, n_class=n_class, pad_idx=pad_idx, n_dec=n_dec, seed=seed, n_head=8,
                                n_query=8, query_num=query_num, mode="train", torch=torch.__version__)),
        "param_names": json.dumps([n for n, _ in names_shapes]),
        "param_shapes": json.dumps([list(s) for _, s in names_shapes]),
        "state_keys": json.dumps(list(model.state_dict().keys())),
        "state_shapes": json.dumps([list(v.shape) for v in model.state_dict().values()]),
        "query": query.numpy(),
        "out_action": out["action"].detach().numpy(), "out_duration": out["duration"].detach().numpy(),
        "out_seg": out["seg"].detach().numpy(), "probe_loss": np.array([float(loss)]),
        "live_names": json.dumps(live),
        "grad_stats": np.stack([G.stats(grads[n]) for n in live]),
        "grad::query_embed.weight": grads["query_embed.weight"].numpy(),
        "grad::fc_seg.weight": grads["fc_seg.weight"].numpy(),
        "grad::input_embed.bias": grads["input_embed.bias"].numpy(),
        "grad::pos_embedding[:S]": grads["pos_embedding"][0, :S].numpy(),
    }
    p0 = {n: torch.from_numpy(synth.fill_value(n, s, j)).requires_grad_(n in live) for j, (n, s) in enumerate(names_shapes)}
    oout, _ = O.forward_proposed(p0, (feats, lab), query, "train", pad_idx, 8, n_dec, 8)
    for k in ("action", "duration", "seg"):
        G.check_close(f"{tag}/out/{k}", oout[k], out[k])
    probe_loss(oout).backward()
    assert sorted(live) == sorted(n for n, q in p0.items() if q.grad is not None)
    for n in live:
        G.check_close(f"{tag}/grad/{n}", p0[n].grad, grads[n], tol=5e-5 * max(1.0, float(grads[n].abs().max())))
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[golden] {tag}: probe loss={float(loss):.6f} live={len(live)} -> {os.path.getsize(path)/1024:.1f} KB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    case("proposed_tiny", H=32, B=2, S=5, n_class=7, n_dec=1, query_num=11, seed=1)
    case("proposed_h128", H=128, B=4, S=16, n_class=17, n_dec=2, query_num=48, seed=10)
