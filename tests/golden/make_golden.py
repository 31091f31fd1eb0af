#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (olivesgatech/R3D at /root/reference) on CPU.

Run in the build container only (the reference does not travel):
    python tests/golden/make_golden.py
What is committed is data: seeds/shape descriptors of the hash-generated inputs and parameters
(oracle/synth.py regenerates them bit-identically anywhere) plus the reference's OUTPUTS.

The only shim is the one SURVEY.md 8(c) describes: CMFuser.generate_cross_attention_mask returns a
Tensor subclass whose .to('cuda') is a no-op (model/futr_safuser_tokenfusion.py:77 hard-codes 'cuda').
Dropout probabilities are set to 0 for the train()-driven case (RNG parity is impossible); all other
cases run the reference in model.eval() state, its de-facto state after the first validate().
While generating, every case is also compared with oracle/futr_oracle.py and the script aborts on a
mismatch, so the committed fixtures certify "oracle == reference" at generation time.
"""
import importlib
import io
import json
import os
import sys
import tempfile
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path[:0] = ["/root/reference", "/root/reference/train"]

from oracle import synth, futr_oracle as O  # noqa: E402

from opts import parser  # noqa: E402  (reference opts.py:2)
import train_proposed_depth as T  # noqa: E402
import utils as RU  # noqa: E402


class _CpuMask(torch.Tensor):
    def to(self, *a, **k):
        return self.as_subclass(torch.Tensor)


M = importlib.import_module("model.futr_safuser_tokenfusion")
_orig = M.CMFuser.__dict__["generate_cross_attention_mask"].__func__
M.CMFuser.generate_cross_attention_mask = staticmethod(lambda sz: _orig(sz).as_subclass(_CpuMask))

torch.set_num_threads(8)


def build_reference(H, n_class, n_dec, n_head=8, n_query=8):
    args = parser.parse_args([])
    args.hidden_dim, args.n_head, args.n_decoder_layer, args.n_query = H, n_head, n_dec, n_query
    pad_idx = n_class + 1
    model = M.FUTR(n_class, H, device=torch.device("cpu"), args=args, src_pad_idx=pad_idx,
                   n_query=n_query, n_head=n_head, num_encoder_layers=args.n_encoder_layer,
                   num_decoder_layers=n_dec)
    names_shapes = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    state = synth.fill_state(names_shapes)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[n]))
    return model, args, pad_idx, names_shapes


def stats(t):
    a = t.detach().double().reshape(-1)
    n = a.numel()
    idx = torch.linspace(0, n - 1, steps=min(n, 16)).long()
    return np.concatenate([[float(a.norm()), float(a.sum()), float(a.abs().sum())],
                           a[:16].numpy() if n >= 16 else np.pad(a.numpy(), (0, 16 - n)),
                           a[idx].numpy() if n >= 16 else np.pad(a[idx].numpy(), (0, 16 - idx.numel()))])


def t_batch(b):
    return [torch.from_numpy(x) for x in b]


def check_close(name, a, b, tol=2e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = float((a - b).abs().max()) if a.numel() else 0.0
    scale = max(1.0, float(b.abs().max()) if b.numel() else 1.0)
    if not err <= tol * scale:
        raise SystemExit(f"ORACLE != REFERENCE at {name}: max abs err {err} (scale {scale})")
    return err


def ref_losses(out, lab, dur, tgt, pad_idx):
    """The reference's own loss composition, by calling its functions in the order of
    train_proposed_depth.py:139-213."""
    crit = torch.nn.MSELoss(reduction="none")
    dur_mask = (dur != pad_idx).long()
    target_dur = dur * dur_mask
    seg = out["seg"]
    B, Tt, C = seg.size()
    l_seg, sc, st, _ = RU.cal_performance(seg.view(-1, C), lab.view(-1), pad_idx, exclude_class_idx=47,
                                          reference=None, target_ref=None)
    act = out["action"]
    B, Tq, C = act.size()
    first = T.get_last_non_padding_labels(lab, pad_idx)
    l_act, ac, at, _ = RU.cal_performance(act.view(-1, C), tgt.contiguous().view(-1), pad_idx,
                                          exclude_class_idx=47, reference=first, target_ref=tgt[:, 0])
    od = RU.normalize_duration(out["duration"], dur_mask)
    l_dur = torch.sum(crit(od, target_dur * dur_mask)) / torch.sum(dur_mask)
    return dict(loss_seg=l_seg, loss_action=l_act, loss_dur=l_dur, loss=l_seg + l_act + l_dur,
                seg_correct=sc, seg_total=st, act_correct=ac, act_total=at)


def case_step(tag, H, B, S, n_class, n_dec, seed, lr=1e-3, wd=5e-3, with_step=True, zero_mean_depth=False):
    """forward (eval state, mode='train') + losses + backward + one AdamW step on the reference."""
    model, args, pad_idx, names_shapes = build_reference(H, n_class, n_dec)
    model.eval()
    batch = t_batch(synth.make_batch(B, S, n_class, pad_idx, seed, zero_mean_depth=zero_mean_depth))
    feats, depth, lab, dur, tgt = batch
    fused_holder = {}
    hook = model.fuser.register_forward_hook(lambda m, i, o: fused_holder.__setitem__("fused", o))
    # capture the selected indices the reference's torch.topk returns
    sel = []
    real_topk = torch.topk

    def spy_topk(*a, **k):
        r = real_topk(*a, **k)
        sel.append(r[1].reshape(-1).clone())
        return r
    torch.topk = spy_topk
    try:
        out = model((feats, lab), depth)               # mode='train' default (futr_safuser_tokenfusion.py:164)
    finally:
        torch.topk = real_topk
        hook.remove()
    res = ref_losses(out, lab, dur, tgt, pad_idx)
    res["loss"].backward()
    fused = fused_holder["fused"].detach()
    fx = {
        "meta": json.dumps(dict(tag=tag, H=H, B=B, S=S, n_class=n_class, pad_idx=pad_idx, n_dec=n_dec, seed=seed,
                                n_head=8, n_query=8, mode="train", lr=lr, wd=wd,
                                zero_mean_depth=zero_mean_depth, torch=torch.__version__)),
        "param_names": json.dumps([n for n, _ in names_shapes]),
        "param_shapes": json.dumps([list(s) for _, s in names_shapes]),
        "out_action": out["action"].detach().numpy(), "out_duration": out["duration"].detach().numpy(),
        "out_seg": out["seg"].detach().numpy(), "fused": fused.numpy(),
        "idx_rgb": np.sort(sel[0].numpy()), "idx_dep": np.sort(sel[1].numpy()),
        "losses": np.array([float(res[k].detach()) for k in ("loss_seg", "loss_action", "loss_dur", "loss")], np.float64),
        "counts": np.array([res[k] for k in ("seg_correct", "seg_total", "act_correct", "act_total")], np.int64),
        "erank_fused": np.array([O.effective_rank(fused.reshape(-1, H))]),
        "erank_svals": torch.linalg.svdvals(fused.reshape(-1, H).double()).numpy(),
    }
    grads = {n: p.grad for n, p in model.named_parameters()}
    live = [n for n, g in grads.items() if g is not None]
    fx["live_names"] = json.dumps(live)
    fx["grad_stats"] = np.stack([stats(grads[n]) for n in live])
    # the full (small) gradients most sensitive to indexing mistakes
    for n in ("query_embed.weight", "fc.weight", "fc_len.weight", "fc_seg.weight", "depth_layernorm.weight",
              "input_embed.bias", "depth_projection.bias", "fuser.norm.weight",
              "transformer.decoder.norm.weight"):
        fx["grad::" + n] = grads[n].numpy()
    fx["grad::pos_embedding[:S]"] = grads["pos_embedding"][0, :S].numpy()
    fx["grad::qkv_qk_absmax"] = np.array([float(grads["fuser.blocks.0.attn.qkv.weight"][:2 * H].abs().max())])

    # ---- oracle cross-check (forward, losses, grads) ---------------------------------------------------
    p0 = {n: torch.from_numpy(synth.fill_value(n, s, j)) for j, (n, s) in enumerate(names_shapes)}
    tr = O.CpuTrainer(p0, pad_idx, n_head=8, n_layers=n_dec, lr=lr, wd=wd)
    ores, oout, oaux = tr.step(batch, apply=False)
    for k in ("action", "duration", "seg"):
        check_close(f"{tag}/out/{k}", oout[k], out[k])
    check_close(f"{tag}/fused", oaux["fused"], fused)
    assert np.array_equal(np.sort(oaux["idx_rgb"].numpy()), fx["idx_rgb"]), "idx_rgb"
    assert np.array_equal(np.sort(oaux["idx_dep"].numpy()), fx["idx_dep"]), "idx_dep"
    for k in ("loss_seg", "loss_action", "loss_dur", "loss"):
        check_close(f"{tag}/{k}", ores[k], res[k])
    for k in ("seg_correct", "seg_total", "act_correct", "act_total"):
        assert ores[k] == res[k], k
    assert sorted(live) == sorted(n for n, q in tr.p.items() if q.grad is not None), "live set"
    for n in live:
        g = grads[n]
        check_close(f"{tag}/grad/{n}", tr.p[n].grad, g, tol=5e-5 * max(1.0, float(g.abs().max())))

    if with_step:
        opt = torch.optim.AdamW(model.parameters(), lr, weight_decay=wd)   # main_darai.py:135
        opt.step()
        post = dict(model.named_parameters())
        fx["post_stats"] = np.stack([stats(post[n]) for n in live])
        dead = [n for n in post if n not in live]
        fx["dead_unchanged"] = np.array([bool(torch.equal(post[n].detach(), p0[n])) for n in dead])
        with torch.no_grad():
            tr.t += 1
            for n in live:
                O.adamw_step(tr.p[n], tr.p[n].grad, tr.m[n], tr.v[n], tr.t, lr, wd)
        # Step-1 AdamW moves every element by lr*g/(|g|+eps): where |g| <~ 1e-6 (e.g. the key bias of
        # an attention, whose true gradient is 0) the update is rounding noise of size <= lr, so only
        # well-conditioned elements are compared tightly; the rest must stay within 2*lr.
        for n in live:
            g = grads[n]
            well = g.abs() > 1e-5
            d = (tr.p[n].detach() - post[n].detach()).abs()
            if well.any() and float(d[well].max()) > 2e-5:
                raise SystemExit(f"ORACLE != REFERENCE at {tag}/post/{n}: {float(d[well].max())}")
            if float(d.max()) > 2.1 * lr:
                raise SystemExit(f"ORACLE != REFERENCE at {tag}/post/{n} (ill-conditioned part): {float(d.max())}")
        fx["post_well_frac"] = np.array([float((grads[n].abs() > 1e-5).float().mean()) for n in live])
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[golden] {tag}: loss={float(res['loss']):.6f} erank={fx['erank_fused'][0]:.4f} "
          f"idx_rgb[:4]={fx['idx_rgb'][:4]} -> {os.path.getsize(path)/1024:.1f} KB")


def case_val(tag, H, S, n_class, n_dec, seed):
    """mode='val' forward, B=1 (train_proposed_depth.py:72): no padding mask, activation-magnitude scores."""
    model, args, pad_idx, names_shapes = build_reference(H, n_class, n_dec)
    model.eval()
    batch = t_batch(synth.make_batch(1, S, n_class, pad_idx, seed, pad_tail=False))
    feats, depth, lab, dur, tgt = batch
    sel = []
    real_topk = torch.topk

    def spy_topk(*a, **k):
        r = real_topk(*a, **k)
        sel.append((a[0].reshape(-1).clone(), r[1].reshape(-1).clone()))
        return r
    torch.topk = spy_topk
    try:
        with torch.no_grad():
            out = model((feats, lab), depth, mode="val")
    finally:
        torch.topk = real_topk
    p0 = {n: torch.from_numpy(synth.fill_value(n, s, j)) for j, (n, s) in enumerate(names_shapes)}
    with torch.no_grad():
        oout, oaux = O.forward(p0, (feats, lab), depth, "val", pad_idx, 8, n_dec)
    for k in ("action", "duration", "seg"):
        check_close(f"{tag}/out/{k}", oout[k], out[k])
    fx = {
        "meta": json.dumps(dict(tag=tag, H=H, B=1, S=S, n_class=n_class, pad_idx=pad_idx, n_dec=n_dec, seed=seed,
                                n_head=8, n_query=8, mode="val", torch=torch.__version__)),
        "param_names": json.dumps([n for n, _ in names_shapes]),
        "param_shapes": json.dumps([list(s) for _, s in names_shapes]),
        "out_action": out["action"].numpy(), "out_duration": out["duration"].numpy(), "out_seg": out["seg"].numpy(),
        "idx_rgb": np.sort(sel[0][1].numpy()), "idx_dep": np.sort(sel[1][1].numpy()),
        "score_rgb": sel[0][0].numpy(), "score_dep": sel[1][0].numpy(),
    }
    assert np.array_equal(np.sort(oaux["idx_rgb"].numpy()), fx["idx_rgb"])
    assert np.array_equal(np.sort(oaux["idx_dep"].numpy()), fx["idx_dep"])
    # margin at the selection boundary (how safe the set is against summation-order noise)
    for nm in ("rgb", "dep"):
        s = np.sort(fx["score_" + nm])
        k = H // 4
        fx["gap_" + nm] = np.array([(s[k] - s[k - 1]) / s[k]])
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[golden] {tag}: gap_rgb={fx['gap_rgb'][0]:.2e} gap_dep={fx['gap_dep'][0]:.2e} "
          f"-> {os.path.getsize(path)/1024:.1f} KB")


def case_train_loop(tag, H, B, S, n_class, n_steps, seed):
    """The reference's own train() (train_proposed_depth.py:110) for one epoch of n_steps batches with
    dropout p=0, torch AdamW (lr 1e-3, wd 5e-3), a no-op scheduler stand-in (pl_bolts is absent), then
    its validate() on one B=1 clip and its checkpoint write."""
    model, args, pad_idx, names_shapes = build_reference(H, n_class, 1)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    args.epochs = 1
    batches = [t_batch(synth.make_batch(B, S, n_class, pad_idx, seed + i)) for i in range(n_steps)]
    val = [t_batch(synth.make_batch(1, S + 3, n_class, pad_idx, seed + 100, pad_tail=False))]
    live_names = None
    step_stats, step_losses = [], []

    class SpyAdamW(torch.optim.AdamW):
        def step(self, closure=None):
            r = super().step(closure)
            step_stats.append(np.stack([stats(p) for n, p in model.named_parameters() if p.grad is not None]))
            return r

    opt = SpyAdamW(model.parameters(), 1e-3, weight_decay=5e-3)
    real_cp = T.cal_performance
    in_val = {"on": False}

    def spy_cp(*a, **k):
        r = real_cp(*a, **k)
        (val_losses if in_val["on"] else step_losses).append(float(r[0]))
        return r
    val_losses = []
    T.cal_performance = spy_cp
    real_validate = T.validate

    def spy_validate(*a, **k):
        in_val["on"] = True
        try:
            return real_validate(*a, **k)
        finally:
            in_val["on"] = False
    T.validate = spy_validate

    class NoSched:
        def step(self):
            pass
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(buf):
        T.train(args, model, batches, opt, NoSched(), torch.nn.MSELoss(reduction="none"), d, pad_idx,
                torch.device("cpu"), val, seed)
        files = sorted(os.listdir(d))
        ck_keys = list(torch.load(os.path.join(d, files[0]), weights_only=True).keys()) if files else []
    T.cal_performance, T.validate = real_cp, real_validate
    live_names = [n for n, p in model.named_parameters() if p.grad is not None]
    log = buf.getvalue()
    # validate() returns (loss, acc, weighted_acc); it is printed: parse nothing, re-run for the numbers
    vres = real_validate(model, val, torch.nn.MSELoss(reduction="none"), pad_idx, torch.device("cpu"))
    fx = {
        "meta": json.dumps(dict(tag=tag, H=H, B=B, S=S, n_class=n_class, pad_idx=pad_idx, n_dec=1, seed=seed,
                                n_steps=n_steps, n_head=8, n_query=8, lr=1e-3, wd=5e-3, val_S=S + 3,
                                torch=torch.__version__)),
        "param_names": json.dumps([n for n, _ in names_shapes]),
        "param_shapes": json.dumps([list(s) for _, s in names_shapes]),
        "live_names": json.dumps(live_names),
        "step_losses": np.array(step_losses, np.float64).reshape(n_steps, 2),      # (seg, action) per step
        "post_stats": np.stack(step_stats),                                          # [n_steps, n_live, 35]
        "val_result": np.array([float(x) for x in vres], np.float64),
        "ckpt_files": json.dumps(files), "ckpt_keys": json.dumps(ck_keys),
        "stdout": json.dumps(log),
    }
    # oracle cross-check of the same loop
    p0 = {n: torch.from_numpy(synth.fill_value(n, s, j)) for j, (n, s) in enumerate(names_shapes)}
    tr = O.CpuTrainer(p0, pad_idx, 8, 1)
    for i, b in enumerate(batches):
        ores, _, _ = tr.step(b)
        check_close(f"{tag}/step{i}/seg", ores["loss_seg"], step_losses[i * 2])
        check_close(f"{tag}/step{i}/act", ores["loss_action"], step_losses[i * 2 + 1])
        ost = np.stack([stats(tr.p[n]) for n in live_names])
        check_close(f"{tag}/step{i}/post", ost[:, :3], step_stats[i][:, :3], tol=2e-5)
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[golden] {tag}: step losses {fx['step_losses'].round(4).tolist()} val={fx['val_result'].round(4).tolist()} "
          f"ckpts={files} -> {os.path.getsize(path)/1024:.1f} KB")


if __name__ == "__main__":
    case_step("step_tiny", H=32, B=2, S=5, n_class=7, n_dec=1, seed=1)
    case_step("step_cfg2", H=128, B=8, S=16, n_class=17, n_dec=1, seed=1)
    case_step("step_cfg2_zm", H=128, B=8, S=16, n_class=17, n_dec=1, seed=10, zero_mean_depth=True)
    case_step("step_k122_dec2", H=64, B=3, S=7, n_class=122, n_dec=2, seed=13452)
    case_val("val_h128", H=128, S=11, n_class=17, n_dec=1, seed=10)
    case_val("val_h64", H=64, S=23, n_class=17, n_dec=1, seed=1)
    case_train_loop("train_loop", H=64, B=8, S=6, n_class=17, n_steps=2, seed=1)
