#!/usr/bin/env python3
"""Generates tests/golden/unsup_*.npz by IMPORTING THE REFERENCE's depth-as-query model
(model/futr_unsupervised_depth.py) on CPU -- build container only; it needs no shim.

Same conventions as make_golden.py: hash-generated inputs / analytic parameters (oracle/synth.py), the module in eval()
state with mode='train' (RNG-free: dropout of the two PositionalEncodings and of the decoder is off), the reference's own
loss functions composed as train/train_proposed_depth.py:139-213 does, backward, one torch AdamW step.  Depth is 4-D
[B, S, 120, 160] as the model's forward unpacks it (futr_unsupervised_depth.py:107; 160*120 pixels :59).  Every value is
cross-checked against oracle/futr_oracle.py: forward_unsup_depth and the script aborts on a mismatch.
mode='val' is not generated: the reference's forward takes the bare feature tensor there (:91) and its validate() passes
a tuple, so validate() crashes in the reference -- documented, not emulated."""
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
import make_golden as G  # noqa: E402   (helpers: ref_losses, stats, check_close, t_batch)

M = importlib.import_module("model.futr_unsupervised_depth")
DEPTH_HW = (120, 160)


def build(H, n_class, n_dec):
    args = parser.parse_args([])
    args.hidden_dim, args.n_head, args.n_decoder_layer, args.n_query = H, 8, n_dec, 8
    pad_idx = n_class + 1
    model = M.FUTR(n_class, H, device=torch.device("cpu"), args=args, src_pad_idx=pad_idx, n_query=8, n_head=8,
                   num_encoder_layers=args.n_encoder_layer, num_decoder_layers=n_dec)
    names_shapes = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    state = synth.fill_state(names_shapes)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[n]))
    return model, pad_idx, names_shapes


def case(tag, H, B, S, n_class, n_dec, seed, lr=1e-3, wd=5e-3):
    model, pad_idx, names_shapes = build(H, n_class, n_dec)
    model.eval()
    batch = G.t_batch(synth.make_batch(B, S, n_class, pad_idx, seed, depth_hw=DEPTH_HW))
    feats, depth, lab, dur, tgt = batch
    depth4 = depth.reshape(B, S, *DEPTH_HW)                       # the forward unpacks 4 dims (:107)
    out = model((feats, lab), depth4)                             # mode='train' default (:85)
    res = G.ref_losses(out, lab, dur, tgt, pad_idx)
    res["loss"].backward()
    grads = {n: p.grad for n, p in model.named_parameters()}
    live = [n for n, g in grads.items() if g is not None]
    fx = {
        "meta": json.dumps(dict(tag=tag, H=H, B=B, S=S, n_class=n_class, pad_idx=pad_idx, n_dec=n_dec, seed=seed, n_head=8,
                                n_query=8, mode="train", lr=lr, wd=wd, depth_hw=list(DEPTH_HW), torch=torch.__version__)),
        "param_names": json.dumps([n for n, _ in names_shapes]),
        "param_shapes": json.dumps([list(s) for _, s in names_shapes]),
        "state_keys": json.dumps(list(model.state_dict().keys())),
        "state_shapes": json.dumps([list(v.shape) for v in model.state_dict().values()]),
        "pos_table_head": model.state_dict()["pos_enc.pos_table"][0, :S].numpy(),
        "out_action": out["action"].detach().numpy(), "out_duration": out["duration"].detach().numpy(),
        "out_seg": out["seg"].detach().numpy(),
        "losses": np.array([float(res[k].detach()) for k in ("loss_seg", "loss_action", "loss_dur", "loss")], np.float64),
        "counts": np.array([res[k] for k in ("seg_correct", "seg_total", "act_correct", "act_total")], np.int64),
        "live_names": json.dumps(live),
        "grad_stats": np.stack([G.stats(grads[n]) for n in live]),
    }
    for n in ("fc.weight", "fc_len.weight", "fc_seg.weight", "depth_layernorm.weight", "input_embed.bias",
              "depth_projection.bias", "transformer.decoder.norm.weight",
              "transformer.decoder.layers.0.self_attn.in_proj_bias",
              "transformer.decoder.layers.0.multihead_attn.in_proj_bias"):
        fx["grad::" + n] = grads[n].numpy()
    fx["grad::pos_embedding[:S]"] = grads["pos_embedding"][0, :S].numpy()

    # ---- oracle cross-check ------------------------------------------------------------------------------------
    p0 = {n: torch.from_numpy(synth.fill_value(n, s, j)) for j, (n, s) in enumerate(names_shapes)}
    tr = O.CpuTrainer(p0, pad_idx, n_head=8, n_layers=n_dec, lr=lr, wd=wd, unsup_depth=True, n_query=8)
    ores, oout, oaux = tr.step([feats, depth4, lab, dur, tgt], apply=False)
    for k in ("action", "duration", "seg"):
        G.check_close(f"{tag}/out/{k}", oout[k], out[k])
    for k in ("loss_seg", "loss_action", "loss_dur", "loss"):
        G.check_close(f"{tag}/{k}", ores[k], res[k])
    for k in ("seg_correct", "seg_total", "act_correct", "act_total"):
        assert ores[k] == res[k], k
    assert sorted(live) == sorted(n for n, q in tr.p.items() if q.grad is not None), \
        (set(live) ^ set(n for n, q in tr.p.items() if q.grad is not None))
    for n in live:
        g = grads[n]
        G.check_close(f"{tag}/grad/{n}", tr.p[n].grad, g, tol=5e-5 * max(1.0, float(g.abs().max())))
    G.check_close(f"{tag}/pos_table", O.sinusoid_table(3000, H), model.state_dict()["pos_enc.pos_table"], tol=0.0)

    opt = torch.optim.AdamW(model.parameters(), lr, weight_decay=wd)   # main_darai.py:135
    opt.step()
    post = dict(model.named_parameters())
    fx["post_stats"] = np.stack([G.stats(post[n]) for n in live])
    dead = [n for n in post if n not in live]
    fx["dead_unchanged"] = np.array([bool(torch.equal(post[n].detach(), p0[n])) for n in dead])
    fx["post_well_frac"] = np.array([float((grads[n].abs() > 1e-5).float().mean()) for n in live])
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[golden] {tag}: loss={float(res['loss']):.6f} live={len(live)} -> {os.path.getsize(path)/1024:.1f} KB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    case("unsup_tiny", H=32, B=2, S=5, n_class=7, n_dec=1, seed=1)
    case("unsup_h128", H=128, B=8, S=16, n_class=17, n_dec=1, seed=1)
    case("unsup_dec2", H=64, B=3, S=11, n_class=17, n_dec=2, seed=13452)
