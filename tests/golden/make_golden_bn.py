#!/usr/bin/env python3
"""Generates tests/golden/bn_*.npz by IMPORTING THE REFERENCE's BN-blend variant
(model/futr_safuser_batchnormalization.py) on CPU -- build container only.  Same conventions and the same single shim as
make_golden.py (the mask's .to('cuda') becomes a no-op).  Dropout probabilities are set to 0 (RNG parity is impossible);
the module runs in train() state so that BatchNorm uses batch statistics and updates its running statistics, then once
more in eval() state on the updated running statistics.  Every value is cross-checked against oracle/futr_oracle.py."""
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
import make_golden as G  # noqa: E402   (helpers: ref_losses, stats, check_close, t_batch; also installs nothing new)


class _CpuMask(torch.Tensor):
    def to(self, *a, **k):
        return self.as_subclass(torch.Tensor)


M = importlib.import_module("model.futr_safuser_batchnormalization")
_orig = M.CMFuser.__dict__["generate_cross_attention_mask"].__func__
M.CMFuser.generate_cross_attention_mask = staticmethod(lambda sz: _orig(sz).as_subclass(_CpuMask))
DEPTH_HW = (120, 160)            # the committed file builds depth_projection for 160*120 pixels (:154)


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
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    return model, pad_idx, names_shapes


def case(tag, H, B, S, n_class, n_dec, seed):
    model, pad_idx, names_shapes = build(H, n_class, n_dec)
    model.train()
    batch = G.t_batch(synth.make_batch(B, S, n_class, pad_idx, seed, depth_hw=DEPTH_HW))
    feats, depth, lab, dur, tgt = batch
    holder = {}
    hook = model.fuser.register_forward_hook(lambda m, i, o: holder.__setitem__("fused", o))
    sel, real_topk = [], torch.topk

    def spy(*a, **k):
        r = real_topk(*a, **k)
        sel.append(r[1].reshape(-1).clone())
        return r
    torch.topk = spy
    try:
        out = model((feats, lab), depth)
    finally:
        torch.topk = real_topk
        hook.remove()
    res = G.ref_losses(out, lab, dur, tgt, pad_idx)
    res["loss"].backward()
    grads = {n: p.grad for n, p in model.named_parameters()}
    live = [n for n, g in grads.items() if g is not None]
    bufs = {n: b.detach().clone() for n, b in model.named_buffers() if "fuser.bn_" in n}
    # eval-state forward on the updated running statistics (validate()'s state)
    model.eval()
    with torch.no_grad():
        out_eval = model((feats, lab), depth)
    fx = {
        "meta": json.dumps(dict(tag=tag, H=H, B=B, S=S, n_class=n_class, pad_idx=pad_idx, n_dec=n_dec, seed=seed, n_head=8,
                                n_query=8, depth_hw=list(DEPTH_HW), variant="bn", torch=torch.__version__)),
        "param_names": json.dumps([n for n, _ in names_shapes]),
        "param_shapes": json.dumps([list(s) for _, s in names_shapes]),
        "buffer_names": json.dumps(list(bufs)),
        "out_action": out["action"].detach().numpy(), "out_duration": out["duration"].detach().numpy(),
        "out_seg": out["seg"].detach().numpy(), "fused": holder["fused"].detach().numpy(),
        "eval_action": out_eval["action"].numpy(), "eval_duration": out_eval["duration"].numpy(),
        "eval_seg": out_eval["seg"].numpy(),
        "idx_rgb": np.sort(sel[0].numpy()), "idx_dep": np.sort(sel[1].numpy()),
        "losses": np.array([float(res[k].detach()) for k in ("loss_seg", "loss_action", "loss_dur", "loss")], np.float64),
        "counts": np.array([res[k] for k in ("seg_correct", "seg_total", "act_correct", "act_total")], np.int64),
        "live_names": json.dumps(live),
        "grad_stats": np.stack([G.stats(grads[n]) for n in live]),
    }
    for n, b in bufs.items():
        fx["buf::" + n] = b.numpy()
    for n in ("fuser.alpha", "fuser.bn_rgb.weight", "fuser.bn_rgb.bias", "fuser.bn_depth.weight", "fuser.bn_depth.bias",
              "depth_layernorm.weight", "input_embed.bias", "fuser.norm.weight"):
        fx["grad::" + n] = grads[n].numpy()
    # ---- oracle cross-check ------------------------------------------------------------------------------------------
    p0 = {n: torch.from_numpy(synth.fill_value(n, s, j)) for j, (n, s) in enumerate(names_shapes)}
    C = H
    st0 = {}
    for pre in ("fuser.bn_rgb.", "fuser.bn_depth."):
        st0[pre + "running_mean"] = torch.zeros(C)
        st0[pre + "running_var"] = torch.ones(C)
        st0[pre + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    tr = O.CpuTrainer(p0, pad_idx, 8, n_dec, bn_state=st0, bn_training=True)
    ores, oout, oaux = tr.step(batch, apply=False)
    for k in ("action", "duration", "seg"):
        G.check_close(f"{tag}/out/{k}", oout[k], out[k])
    G.check_close(f"{tag}/fused", oaux["fused"], holder["fused"])
    assert np.array_equal(np.sort(oaux["idx_rgb"].numpy()), fx["idx_rgb"]) and np.array_equal(np.sort(oaux["idx_dep"].numpy()), fx["idx_dep"])
    for k in ("loss_seg", "loss_action", "loss_dur", "loss"):
        G.check_close(f"{tag}/{k}", ores[k], res[k])
    assert sorted(live) == sorted(n for n, q in tr.p.items() if q.grad is not None), "live set"
    for n in live:
        g = grads[n]
        G.check_close(f"{tag}/grad/{n}", tr.p[n].grad, g, tol=5e-5 * max(1.0, float(g.abs().max())))
    for n, b in bufs.items():
        G.check_close(f"{tag}/buffer/{n}", tr.bn_state[n].float(), b.float())
    with torch.no_grad():
        eo, _ = O.forward(tr.p, (feats, lab), depth, "train", pad_idx, 8, n_dec, bn_state=tr.bn_state, bn_training=False)
    for k in ("action", "duration", "seg"):
        G.check_close(f"{tag}/eval/{k}", eo[k], out_eval[k])
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[golden-bn] {tag}: loss={float(res['loss']):.6f} idx_rgb={fx['idx_rgb'][:6]} live={len(live)} -> {os.path.getsize(path)/1024:.1f} KB")


if __name__ == "__main__":
    case("bn_tiny", 64, 2, 6, 17, 1, 5)
    case("bn_cfg2", 128, 8, 16, 17, 1, 9)
