"""End-to-end parity of the HIP training step (through the C ABI) against the oracle on the same seeded inputs and
against the golden fixtures generated from the imported reference.  Tolerance: 1e-3 relative (fp32), the bound
BASELINE.json's north_star states; selected token indices bit-exact."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import futr_oracle as O  # noqa: E402
from tests.helpers import load_fixture, fixture_params, fixture_batch, stats, assert_close  # noqa: E402

RTOL = 1e-3


def build_model(fx):
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    m = fx["meta"]
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    model = FUTR(m["n_class"], m["H"], m["pad_idx"], torch.device("cuda"), args, n_query=m["n_query"], n_head=m["n_head"],
                 num_encoder_layers=2, num_decoder_layers=m["n_dec"])
    sd = fixture_params(fx)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("pos_table" in k for k in missing.missing_keys)
    return model.to("cuda")


def close_rel(a, b, what, rtol=RTOL):
    """max-abs error relative to the tensor's scale (element-wise rtol is meaningless for values near 0)"""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-5)
    err = float((a - b).abs().max())
    assert err <= rtol * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


@pytest.mark.parametrize("tag", ["step_tiny", "step_cfg2", "step_cfg2_zm", "step_k122_dec2"])
def test_step_parity(tag, oracle_lib):
    fx = load_fixture(tag)
    m = fx["meta"]
    batch = fixture_batch(fx)
    feats, depth, lab, dur, tgt = batch
    # oracle on CPU
    tr = O.CpuTrainer(fixture_params(fx), m["pad_idx"], m["n_head"], m["n_dec"], m["lr"], m["wd"])
    ores, oout, oaux = tr.step(batch, apply=False)
    # HIP
    model = build_model(fx).eval()            # eval(): dropout off, the parity state
    eng = model.engine()
    d = [t.cuda() for t in batch]
    out = eng.forward(d[0], d[1], d[2], "train", training=False)
    torch.cuda.synchronize()
    for k, fk in (("action", "out_action"), ("duration", "out_duration"), ("seg", "out_seg")):
        close_rel(out[k], oout[k].detach(), f"{tag}/{k} vs oracle")
        close_rel(out[k], fx[fk], f"{tag}/{k} vs reference fixture")
    w = eng.last["w"]
    close_rel(w.fused.view(m["B"], m["S"], m["H"]), fx["fused"], f"{tag}/fused")
    assert np.array_equal(eng.last["idx"][0].cpu().numpy(), fx["idx_rgb"])
    assert np.array_equal(eng.last["idx"][1].cpu().numpy(), fx["idx_dep"])
    loss, counts = eng.losses(d[2], d[4], d[3])
    eng.backward()
    torch.cuda.synchronize()
    assert_close(loss.cpu(), fx["losses"], RTOL, 1e-6, f"{tag}/losses vs fixture")
    assert counts.cpu().tolist() == fx["counts"].tolist()
    live = fx["live_names"]
    for n in live:
        g = eng.arena.g(n)
        close_rel(g, tr.p[n].grad, f"{tag}/grad {n}", rtol=2e-3)
    gs = np.stack([stats(eng.arena.g(n)) for n in live])
    ref = fx["grad_stats"]
    assert bool((np.abs(gs[:, 0] - ref[:, 0]) <= 2e-3 * ref[:, 0] + 1e-7).all()), "grad norms vs reference fixture"
    assert float(eng.arena.g("fuser.blocks.0.attn.qkv.weight")[:2 * m["H"]].abs().max()) == 0.0
    # one fused AdamW step against the oracle's step on ITS gradients (well-conditioned elements only, see make_golden)
    eng.adamw(m["lr"], m["wd"])
    torch.cuda.synchronize()
    tr.t += 1
    with torch.no_grad():
        for n in live:
            gref = tr.p[n].grad.clone()
            O.adamw_step(tr.p[n], gref, tr.m[n], tr.v[n], tr.t, m["lr"], m["wd"])
            well = gref.abs() > max(1e-4 * float(gref.abs().max()), 1e-6)
            dlt = (eng.arena.p(n).cpu() - tr.p[n].detach()).abs()
            assert float(dlt.max()) <= 2.1 * m["lr"], n
            if well.any():
                assert float(dlt[well].max()) <= 5e-5, (n, float(dlt[well].max()))
    for i, n in enumerate(fx["param_names"]):
        if n not in live:       # dead parameters untouched by the optimiser
            assert torch.equal(dict(model.named_parameters())[n].cpu(), fixture_params(fx)[n]), n


@pytest.mark.parametrize("tag", ["val_h128", "val_h64"])
def test_val_mode_parity(tag, oracle_lib):
    fx = load_fixture(tag)
    m = fx["meta"]
    feats, depth, lab, dur, tgt = fixture_batch(fx, pad_tail=False)
    model = build_model(fx).eval()
    with torch.no_grad():
        out = model((feats.cuda(), lab.cuda()), depth.cuda(), mode="val")
    torch.cuda.synchronize()
    for k, fk in (("action", "out_action"), ("duration", "out_duration"), ("seg", "out_seg")):
        close_rel(out[k], fx[fk], f"{tag}/{k}")
    eng = model.engine()
    assert np.array_equal(eng.last["idx"][0].cpu().numpy(), fx["idx_rgb"])       # bit-exact selection
    assert np.array_equal(eng.last["idx"][1].cpu().numpy(), fx["idx_dep"])


def test_autograd_bridge_matches_fused_backward(oracle_lib):
    """The drop-in route: torch losses on the module outputs + .backward() (what the reference's own train() does)."""
    fx = load_fixture("step_tiny")
    m = fx["meta"]
    batch = fixture_batch(fx)
    model = build_model(fx).eval()
    d = [t.cuda() for t in batch]
    out = model((d[0], d[2]), d[1])
    res = O.losses(out, d[2].cpu().cuda(), d[3], d[4], m["pad_idx"]) if False else None
    # use plain torch ops for the loss here (any differentiable function of the outputs must work)
    loss = (out["seg"] ** 2).mean() + out["action"].sum() * 0.01 + (out["duration"] * 0.1).exp().mean()
    loss.backward()
    torch.cuda.synchronize()
    tr = O.CpuTrainer(fixture_params(fx), m["pad_idx"], m["n_head"], m["n_dec"])
    oout, _ = O.forward(tr.p, (batch[0], batch[2]), batch[1], "train", m["pad_idx"], m["n_head"], m["n_dec"])
    ol = (oout["seg"] ** 2).mean() + oout["action"].sum() * 0.01 + (oout["duration"] * 0.1).exp().mean()
    ol.backward()
    for n, p in model.named_parameters():
        if tr.p[n].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            close_rel(p.grad, tr.p[n].grad, f"bridge grad {n}", rtol=2e-3)


def test_fused_train_steps_track_oracle(oracle_lib):
    """Several fused steps (fwd + losses + bwd + AdamW, no host sync) against the oracle loop and the reference's
    own train() capture (tests/golden/train_loop.npz)."""
    fx = load_fixture("train_loop")
    m = fx["meta"]
    model = build_model(fx).eval()
    eng = model.engine()
    tr = O.CpuTrainer(fixture_params(fx), m["pad_idx"], m["n_head"], 1, m["lr"], m["wd"])
    for i in range(m["n_steps"]):
        batch = fixture_batch(fx, seed=m["seed"] + i)
        d = [t.cuda() for t in batch]
        loss, counts = eng.train_step(d[0], d[1], d[2], d[3], d[4], m["lr"], m["wd"], training=False)
        torch.cuda.synchronize()
        ores, _, _ = tr.step(batch)
        assert_close(loss[:2].cpu(), fx["step_losses"][i], 2e-3, 1e-5, f"step {i} losses vs reference train()")
        assert_close(loss[3].cpu(), float(ores["loss"]), 2e-3, 1e-5, f"step {i} total vs oracle")
        ps = np.array([float(eng.arena.p(n).double().norm()) for n in fx["live_names"]])
        assert_close(ps, fx["post_stats"][i][:, 0], 2e-3, 2e-3, f"post-step norms {i}")


def test_dropout_training_mode_statistics():
    """model.train(): dropout masks are active, change every step, and the step still runs end to end."""
    fx = load_fixture("step_tiny")
    m = fx["meta"]
    model = build_model(fx).train()
    eng = model.engine()
    d = [t.cuda() for t in fixture_batch(fx)]
    o1 = {k: v.clone() for k, v in eng.forward(d[0], d[1], d[2], "train", training=True).items()}
    keep = float(eng.last["w"].drop_pool.float().mean())
    eng.drop_offset.add_(1)
    o2 = {k: v.clone() for k, v in eng.forward(d[0], d[1], d[2], "train", training=True).items()}
    torch.cuda.synchronize()
    assert 0.85 < keep < 0.95
    assert float((o1["seg"] - o2["seg"]).abs().max()) > 1e-4
    loss, _ = eng.train_step(d[0], d[1], d[2], d[3], d[4], 1e-3, 5e-3, training=True)
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all()


@pytest.mark.parametrize("tag,training", [("step_tiny", False), ("step_cfg2", False), ("step_k122_dec2", False), ("step_cfg2", True)])
def test_fused_decoder_equals_composed_decoder(tag, training):
    """decoder.hip (one workgroup per clip) against the same layer composed from GEMM / attention / LayerNorm launches,
    with identical dropout masks when training."""
    fx = load_fixture(tag)
    m = fx["meta"]
    model = build_model(fx)
    model.train(training)
    eng = model.engine()
    d = [t.cuda() for t in fixture_batch(fx)]
    outs, acts = [], []
    for fused in (False, True):
        eng.use_fused_decoder = fused
        out = eng.forward(d[0], d[1], d[2], "train", training=training)      # same drop_offset -> same masks
        torch.cuda.synchronize()
        w = eng.last["w"]
        outs.append({k: v.clone() for k, v in out.items()})
        acts.append([{k: v.clone() for k, v in layer.items()} for layer in w.layers] + [dict(tgtF=w.tgtF.clone())])
    from r3d_amd import ops
    assert ops.decoder_fused_supported(m["S"], m["n_query"], m["H"], m["n_head"])
    for k in outs[0]:
        close_rel(outs[1][k], outs[0][k], f"{tag}/{k} fused vs composed", rtol=2e-5)
    for la, lb in zip(acts[0], acts[1]):
        for k in la:
            close_rel(lb[k], la[k], f"{tag}/act {k}", rtol=5e-5)


@pytest.mark.parametrize("B,S,H,K,n_dec", [(8, 64, 512, 17, 1), (8, 16, 1024, 17, 1), (16, 16, 1024, 17, 1), (8, 32, 128, 17, 1),
                                           (16, 16, 256, 122, 2)])
def test_step_parity_baseline_sizes(B, S, H, K, n_dec, oracle_lib):
    """BASELINE.json's other configurations at their per-GPU shapes (cfg4: B=8,S=64,H=512; cfg5: H=1024; cfg3: S=32)
    and an NTU-sized head: one training step against the oracle on the same hash-filled parameters and inputs.
    No reference fixture exists at these sizes (the reference is not on the GPU box); the oracle is pinned to the
    reference at H=64/128 by tests/golden."""
    from oracle import synth
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    pad = K + 1
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    model = FUTR(K, H, pad, torch.device("cuda"), args, n_query=8, n_head=8, num_encoder_layers=2, num_decoder_layers=n_dec)
    names = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    params = {n: torch.from_numpy(v) for n, v in synth.fill_state(names).items()}
    model.load_state_dict(params, strict=False)
    model = model.to("cuda").eval()
    batch = [torch.from_numpy(x) for x in synth.make_batch(B, S, K, pad, 77)]
    tr = O.CpuTrainer(params, pad, 8, n_dec, 1e-3, 5e-3)
    ores, oout, oaux = tr.step(batch, apply=False)
    eng = model.engine()
    d = [t.cuda() for t in batch]
    out = eng.forward(d[0], d[1], d[2], "train", training=False)
    loss, counts = eng.losses(d[2], d[4], d[3])
    eng.backward()
    torch.cuda.synchronize()
    for k in ("action", "duration", "seg"):
        close_rel(out[k], oout[k].detach(), f"H{H}/{k}")
    close_rel(eng.last["w"].fused.view(B, S, H), oaux["fused"].detach(), f"H{H}/fused")
    assert_close(loss[3].cpu(), float(ores["loss"]), 1e-3, 1e-6, "total loss")
    assert torch.equal(eng.last["idx"][0].cpu(), oaux["idx_rgb"]) and torch.equal(eng.last["idx"][1].cpu(), oaux["idx_dep"])
    # ReLU units of the decoder FFN whose pre-activation is within rounding of zero in the oracle: the fp32 product here may
    # land on the other side of the kink, which flips that unit's contribution (its row of linear1's gradient entirely,
    # everything upstream by the unit's share).  They are identified BY CONSTRUCTION from the oracle's pre-activations
    # (|u| <= 2e-6 max|u|: the size of the fp32 accumulation error of a K = H dot product), must be few, and only they get
    # a different treatment: their rows of linear1's gradient are excluded, and -- only if such a unit exists -- the
    # other gradients get the bound of one flipped unit (4e-3) instead of 2e-3.
    kink_units = set()
    for l, u in enumerate(oaux["ffn_pre"]):
        near = (u.abs() <= 2e-6 * float(u.abs().max())).reshape(-1, u.shape[-1])
        kink_units |= {(l, int(j)) for j in near.any(dim=0).nonzero().flatten()}
    assert len(kink_units) <= 4, f"H{H}: {len(kink_units)} FFN units on the ReLU kink"
    rtol = 2e-3 if not kink_units else 4e-3
    for n, p in tr.p.items():
        if p.grad is None:
            continue
        g, r = eng.arena.g(n).cpu(), p.grad
        for (l, j) in kink_units:
            if n in (f"transformer.decoder.layers.{l}.linear1.weight", f"transformer.decoder.layers.{l}.linear1.bias"):
                keep = torch.ones(r.shape[0], dtype=torch.bool)
                keep[j] = False
                g, r = g[keep], r[keep]
            elif n == f"transformer.decoder.layers.{l}.linear2.weight":
                keep = torch.ones(r.shape[1], dtype=torch.bool)
                keep[j] = False
                g, r = g[:, keep], r[:, keep]
        close_rel(g, r, f"H{H}/grad {n}" + (f" (kink units {sorted(kink_units)} excluded)" if kink_units else ""), rtol=rtol)


@pytest.mark.parametrize("tag,training", [("step_tiny", False), ("step_cfg2", False), ("step_cfg2", True), ("step_k122_dec2", True)])
def test_fused_embedding_seam_equals_composed(tag, training):
    """embed.hip (slab sums + LN + exchange + norm1 in one launch per direction) against the composed launches: same
    activations, same gradients (identical dropout masks when training)."""
    fx = load_fixture(tag)
    model = build_model(fx)
    model.train(training)
    eng = model.engine()
    d = [t.cuda() for t in fixture_batch(fx)]
    res = []
    for seam in (False, True):
        eng.use_fused_embed = seam
        eng.forward(d[0], d[1], d[2], "train", training=training)          # same drop_offset -> same masks
        assert eng.last["seam"] == seam
        eng.losses(d[2], d[4], d[3])
        eng.backward()
        torch.cuda.synchronize()
        w = eng.last["w"]
        res.append(dict(acts={k: getattr(w, k).clone() for k in ("rgb", "dep", "dep_pre", "mean_d", "rstd_d", "x0", "h1",
                                                                 "m1", "r1", "fused", "d_rgb_pre", "d_dep_pre")},
                        grads=eng.arena.grads.clone(), loss=w.loss.clone()))
    for k in res[0]["acts"]:
        close_rel(res[1]["acts"][k], res[0]["acts"][k], f"{tag}/seam {k}", rtol=2e-5)
    close_rel(res[1]["loss"], res[0]["loss"], "loss", rtol=1e-5)
    a = eng.arena
    for n in a.live_names:
        o, k, _ = a.offsets[n]
        close_rel(res[1]["grads"][o:o + k], res[0]["grads"][o:o + k], f"{tag}/seam grad {n}", rtol=5e-4)


@pytest.mark.parametrize("tag,training", [("step_cfg2", False), ("step_cfg2", True), ("step_cfg2_zm", True)])
def test_fuser_chain_kernel_equals_composed_launches(tag, training):
    """csrc/fuser_chain.hip (the fuser block's row-local chain + the query self-attention sub-layer, one launch per
    direction) against the grouped GEMM / gemm_ln launches it replaces: every stored activation, the outputs, the losses and
    every gradient (identical dropout masks when training)."""
    fx = load_fixture(tag)
    model = build_model(fx)
    model.train(training)
    eng = model.engine()
    d = [t.cuda() for t in fixture_batch(fx)]
    res = []
    names = ("vsw", "x1", "h2", "m2", "r2", "u", "f1", "x3", "mf", "rf", "fused", "seg")
    lnames = ("cakv", "sa_qkv", "p_sa", "sa_o", "t1_pre", "t1", "m1", "r1", "caq")
    gnames = ("d_x3", "d_u", "d_h2", "d_x1", "d_v", "d_h1", "d_rgb_pre", "d_dep_pre", "d_fused")
    for chain in (False, True):
        eng.use_fuser_chain = chain
        eng.forward(d[0], d[1], d[2], "train", training=training)          # same drop_offset -> same masks
        w = eng.last["w"]
        assert eng._chain_ok(w) == chain
        eng.losses(d[2], d[4], d[3])
        eng.backward()
        torch.cuda.synchronize()
        acts = {k: getattr(w, k).clone() for k in names + gnames}
        acts.update({k: w.layers[0][k].clone() for k in lnames})
        acts.update({"g_" + k: w.glayers[0][k].clone() for k in ("caqin", "sap", "sao", "saqkv", "sain")})
        res.append(dict(acts=acts, grads=eng.arena.grads.clone(), loss=w.loss.clone()))
    for k in res[0]["acts"]:
        close_rel(res[1]["acts"][k], res[0]["acts"][k], f"{tag}/chain {k}", rtol=5e-5)
    close_rel(res[1]["loss"], res[0]["loss"], "loss", rtol=1e-5)
    a = eng.arena
    for n in a.live_names:
        o, k, _ = a.offsets[n]
        close_rel(res[1]["grads"][o:o + k], res[0]["grads"][o:o + k], f"{tag}/chain grad {n}", rtol=5e-4)


@pytest.mark.parametrize("tag,training,defer", [("step_cfg2", False, False), ("step_cfg2", True, False), ("step_cfg2", True, True),
                                                 ("step_cfg2_zm", False, True)])
def test_decoder_chain_kernel_equals_composed_launches(tag, training, defer):
    """csrc/decoder_chain.hip (the decoder layer's query side: cross-attention core, out_proj + norm2, FFN -- and with
    defer_tail the tail, the losses and the backward half in the same launch) against the launches it replaces: stored
    activations, outputs, losses, input gradients and every parameter gradient (identical dropout masks when training)."""
    fx = load_fixture(tag)
    model = build_model(fx)
    model.train(training)
    eng = model.engine()
    d = [t.cuda() for t in fixture_batch(fx)]
    res = []
    lnames = ("p_ca", "ca_o", "t2_pre", "t2", "m2", "r2", "ff1", "t3_pre", "t3", "m3", "r3")
    gnames = ("t3pre", "ff2", "ff1", "t2pre", "cap", "cao", "caq", "cakv")
    for chain in (False, True):
        eng.use_decoder_chain = chain
        eng.defer_tail = defer and chain
        eng.forward(d[0], d[1], d[2], "train", training=training)          # same drop_offset -> same masks
        w = eng.last["w"]
        assert bool(getattr(w, "_dec_deferred", False)) == (defer and chain)
        eng.losses(d[2], d[4], d[3])
        eng.backward()
        torch.cuda.synchronize()
        acts = {k: w.layers[0][k].clone() for k in lnames}
        acts.update({"g_" + k: w.glayers[0][k].clone() for k in gnames})
        acts.update(actdur=w.actdur.clone(), tgtF=w.tgtF.clone(), d_seg=w.d_seg.clone(), d_actdur=w.d_actdur.clone())
        res.append(dict(acts=acts, grads=eng.arena.grads.clone(), loss=w.loss.clone(), counts=w.counts.clone()))
    eng.defer_tail = False
    for k in res[0]["acts"]:
        close_rel(res[1]["acts"][k], res[0]["acts"][k], f"{tag}/decoder chain {k}", rtol=5e-5)
    close_rel(res[1]["loss"], res[0]["loss"], "loss", rtol=1e-5)
    assert torch.equal(res[1]["counts"], res[0]["counts"])
    a = eng.arena
    for n in a.live_names:
        o, k, _ = a.offsets[n]
        close_rel(res[1]["grads"][o:o + k], res[0]["grads"][o:o + k], f"{tag}/decoder chain grad {n}", rtol=5e-4)


def test_adamw_in_weight_gradient_epilogue_equals_flat_adamw():
    """backward(fused_adamw=...) updates depth_projection.weight inside its weight-gradient GEMM; parameters and both
    moments must match the flat AdamW launch on the stored gradient up to the well-conditioned criterion (a rounding
    difference flips the sign of noise-level gradients and Adam turns a sign into +-lr, see test_step_parity)."""
    fx = load_fixture("step_cfg2")
    m = fx["meta"]
    d = [t.cuda() for t in fixture_batch(fx)]
    states = []
    for fused in (False, True):
        model = build_model(fx).eval()
        eng = model.engine()
        snaps = []
        for _ in range(2):
            eng.forward(d[0], d[1], d[2], "train", training=False)
            eng.losses(d[2], d[4], d[3], tick=True)
            eng.backward(fused_adamw=dict(lr=m["lr"], weight_decay=m["wd"]) if fused else None)
            eng.adamw(m["lr"], m["wd"], ticked=True, skip_depth=fused)
            torch.cuda.synchronize()
            a = eng.arena
            snaps.append((a.params[:a.n_live].clone(), a.exp_avg.clone(), a.exp_avg_sq.clone()))
        states.append(snaps)
    # the two routes use different GEMM kernels for the gradient (persistent panels vs 64x64 tiles + epilogue): the
    # moments agree to rounding, the parameters under Adam's usual sign sensitivity where |g| is at noise level
    close_rel(states[1][0][1], states[0][0][1], "fused AdamW step 1 exp_avg", rtol=1e-5)
    close_rel(states[1][0][2], states[0][0][2], "fused AdamW step 1 exp_avg_sq", rtol=1e-5)
    for k in (0, 1):
        dlt = (states[0][k][0] - states[1][k][0]).abs()
        assert float(dlt.max()) <= 2.1 * m["lr"] * (k + 1)
        assert float((dlt <= 1e-5 * (1 + states[0][k][0].abs())).double().mean()) > 0.99


def test_wide_model_parallel_branches_change_no_bit():
    """Hidden 512 on one rank: train_step() forks the query self-attention branch and the parameter-gradient tail to second
    streams (auto_side_stream) and updates depth_projection.weight in its weight-gradient kernel; against the single-stream
    sequence the parameters, moments and losses after three steps must be bit-identical (no arithmetic changes, only order of
    independent launches)."""
    from oracle import synth
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    K, pad, B, S, H = 17, 18, 4, 64, 512
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    outs = []
    for auto in (False, True):
        model = FUTR(K, H, pad, torch.device("cuda"), args, n_query=8, n_head=8, num_encoder_layers=2, num_decoder_layers=1)
        names = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
        model.load_state_dict({n: torch.from_numpy(v) for n, v in synth.fill_state(names).items()}, strict=False)
        model = model.to("cuda").train()
        eng = model.engine()
        eng.auto_side_stream = auto
        d = [torch.from_numpy(x).cuda() for x in synth.make_batch(B, S, K, pad, 77)]
        for _ in range(3):
            loss, _ = eng.train_step(d[0], d[1], d[2], d[3], d[4], 1e-3, 5e-3, training=True)
        torch.cuda.synchronize()
        assert not eng._tail_pending
        outs.append((eng.arena.params.clone(), eng.arena.exp_avg.clone(), eng.last["w"].loss.clone()))
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("H", [256, 512])
def test_wide_tail_losses_launch_equals_the_three_launches(H):
    """Hidden 129 .. 512: decoder tail forward + the three losses + tail backward as ONE launch (tail_losses_wide_kernel: head
    weights and LayerNorm partials in dynamic LDS) against the three kernels it replaces -- losses, counters, every gradient."""
    from oracle import synth
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    from r3d_amd import ops
    K, pad, B, S = 17, 18, 4, 32
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    model = FUTR(K, H, pad, torch.device("cuda"), args, n_query=8, n_head=8, num_encoder_layers=2, num_decoder_layers=1)
    names = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    model.load_state_dict({n: torch.from_numpy(v) for n, v in synth.fill_state(names).items()}, strict=False)
    model = model.to("cuda").eval()
    eng = model.engine()
    assert ops.tail_losses_supported(H, K + 1, 8, B * 8)
    d = [torch.from_numpy(x).cuda() for x in synth.make_batch(B, S, K, pad, 77)]
    res, calls = [], []
    real = ops.decoder_tail_losses
    ops.decoder_tail_losses = lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1]
    try:
        for fused in (False, True):
            eng.use_fused_tail = fused
            eng.defer_tail = fused
            eng.forward(d[0], d[1], d[2], "train", training=False)
            loss, counts = eng.losses(d[2], d[4], d[3])
            eng.backward()
            torch.cuda.synchronize()
            res.append((loss.clone(), counts.clone(), eng.arena.grads.clone()))
            assert len(calls) == (1 if fused else 0), "the one-launch tail must run exactly in the fused pass"
    finally:
        ops.decoder_tail_losses = real
        eng.defer_tail = False
    close_rel(res[1][0], res[0][0], "loss", rtol=1e-5)
    assert torch.equal(res[1][1], res[0][1])
    a = eng.arena
    for n in a.live_names:
        o, k, _ = a.offsets[n]
        close_rel(res[1][2][o:o + k], res[0][2][o:o + k], f"H{H} grad {n}", rtol=5e-4)


@pytest.mark.parametrize("B,S,H", [(4, 64, 256), (8, 32, 128)])
def test_depth_adamw_in_the_bf16x3_weight_gradient_kernel(B, S, H):
    """train_step() updates depth_projection.weight inside its weight-gradient kernel where that product runs on the tiled
    bf16x3 kernel (more than 128 token rows or hidden units: engine.depth_adamw_fusable()); two steps must leave the same
    parameters and moments as the flat AdamW launch on the stored gradient (same kernel, same products: agreement to rounding)."""
    from oracle import synth
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    K, pad = 17, 18
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    states = []
    for fuse in (False, True):
        model = FUTR(K, H, pad, torch.device("cuda"), args, n_query=8, n_head=8, num_encoder_layers=2, num_decoder_layers=1)
        names = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
        model.load_state_dict({n: torch.from_numpy(v) for n, v in synth.fill_state(names).items()}, strict=False)
        model = model.to("cuda").eval()
        eng = model.engine()
        eng.fuse_depth_adamw = fuse
        d = [torch.from_numpy(x).cuda() for x in synth.make_batch(B, S, K, pad, 77)]
        for _ in range(2):
            eng.train_step(d[0], d[1], d[2], d[3], d[4], 1e-3, 5e-3, training=False)
        torch.cuda.synchronize()
        assert eng.depth_adamw_fusable() == fuse
        a = eng.arena
        states.append((a.params[:a.n_live].clone(), a.exp_avg.clone(), a.exp_avg_sq.clone()))
    close_rel(states[1][1], states[0][1], "exp_avg", rtol=1e-5)
    close_rel(states[1][2], states[0][2], "exp_avg_sq", rtol=1e-5)
    dlt = (states[0][0] - states[1][0]).abs()
    assert float(dlt.max()) <= 2.1 * 1e-3 * 2                       # (Adam turns a noise-level sign difference into +-lr per step)
    assert float((dlt <= 1e-5 * (1 + states[0][0].abs())).double().mean()) > 0.99


@pytest.mark.parametrize("flags", [dict(use_fused_tail=False), dict(use_paired_launches=False),
                                   dict(use_fused_tail=False, use_paired_launches=False),
                                   dict(use_fused_tail=False, use_paired_launches=False, use_fused_embed=False),
                                   dict(use_side_stream=True), dict(use_fused_decoder=True), dict(use_gemm_ln=False),
                                   dict(use_gemm_ln=False, use_paired_launches=False), dict(fold_rowsums=False), dict(ride_attention=False),
                                   dict(ride_attention_bwd=True), dict(split_k4h=True), dict(use_fuser_chain=False),
                                   dict(use_decoder_chain=False), dict(use_decoder_chain=False, use_fuser_chain=False),
                                   dict(use_decoder_chain=True, use_fused_tail=False), dict(chain_bf3=False),
                                   dict(overlap_planes=True), dict(overlap_param_tail=True), dict(ride_planes=False), dict(pair_embeddings=False)])
@pytest.mark.parametrize("tag,training", [("step_cfg2", True), ("step_k122_dec2", False)])
def test_launch_fusion_paths_agree(tag, training, flags):
    """Every launch-fusion switch of the engine (decoder tail kernel, paired GEMM launches, embedding seam, side stream,
    fused decoder layer) must leave losses and gradients unchanged (identical dropout masks when training)."""
    fx = load_fixture(tag)
    model = build_model(fx)
    model.train(training)
    eng = model.engine()
    d = [t.cuda() for t in fixture_batch(fx)]
    res = []
    for alt in (False, True):
        if not alt:
            defaults = {k: getattr(eng, k) for k in flags}
        else:
            for k, v in flags.items():
                setattr(eng, k, v)
        eng.forward(d[0], d[1], d[2], "train", training=training)          # same drop_offset -> same masks
        eng.losses(d[2], d[4], d[3])
        eng.backward()
        torch.cuda.synchronize()
        res.append((eng.last["w"].loss.clone(), eng.arena.grads.clone()))
        if alt:
            for k, v in defaults.items():
                setattr(eng, k, v)
    close_rel(res[1][0], res[0][0], "loss", rtol=1e-5)
    a = eng.arena
    for n in a.live_names:
        o, k, _ = a.offsets[n]
        close_rel(res[1][1][o:o + k], res[0][1][o:o + k], f"{tag}/{flags} grad {n}", rtol=5e-4)


def test_dropout_prefill_in_adamw_launch_gives_the_same_masks():
    """adamw(prefill_dropout=True) fills the next step's dropout pool inside the optimiser launch; three training steps
    (dropout on) must be bit-identical to generating the masks at the head of each step."""
    fx = load_fixture("step_cfg2")
    m = fx["meta"]
    d = [t.cuda() for t in fixture_batch(fx)]
    outs = []
    for prefill in (False, True):
        model = build_model(fx).train()
        eng = model.engine()
        for _ in range(3):
            eng.forward(d[0], d[1], d[2], "train", training=True)
            eng.losses(d[2], d[4], d[3], tick=True)
            eng.backward()
            eng.adamw(m["lr"], m["wd"], ticked=True, prefill_dropout=prefill)
        torch.cuda.synchronize()
        outs.append((eng.arena.params.clone(), eng.last["w"].drop_pool.clone(), eng.last["w"].loss.clone()))
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[1][0])


def test_parallel_graph_branches_change_no_bit():
    """overlap_planes / overlap_param_tail move the weight re-split, the grouped weight-gradient launch and the small
    bucket's AdamW to a second stream (fork / join by events, as under graph capture): three training steps through
    train_step() must leave bit-identical parameters and losses."""
    fx = load_fixture("step_cfg2")
    m = fx["meta"]
    d = [t.cuda() for t in fixture_batch(fx)]
    outs = []
    for ov in (False, True):
        model = build_model(fx).train()
        eng = model.engine()
        eng.overlap_planes = eng.overlap_param_tail = ov
        for _ in range(3):
            loss, _ = eng.train_step(d[0], d[1], d[2], d[3], d[4], m["lr"], m["wd"], training=True)
        torch.cuda.synchronize()
        assert not eng._tail_pending and not eng._planes_forked
        outs.append((eng.arena.params.clone(), eng.arena.exp_avg.clone(), eng.last["w"].loss.clone()))
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("tag,paired", [("step_cfg2", True), ("step_cfg2", False), ("step_tiny", True)])
def test_effective_rank_penalty_gradients(tag, paired, oracle_lib):
    """erank_weight: total loss = L - w * erank(fused tokens).  Every gradient against autograd through
    torch.linalg.svdvals on the oracle (the reference has no such term: SURVEY F1, build-side definition)."""
    lam = 0.05
    fx = load_fixture(tag)
    m = fx["meta"]
    batch = fixture_batch(fx)
    tr = O.CpuTrainer(fixture_params(fx), m["pad_idx"], m["n_head"], m["n_dec"])
    out, aux = O.forward(tr.p, (batch[0], batch[2]), batch[1], "train", m["pad_idx"], m["n_head"], m["n_dec"])
    res = O.losses(out, batch[2], batch[3], batch[4], m["pad_idx"])
    er = O.effective_rank_torch(aux["fused"].reshape(-1, m["H"]).double())
    (res["loss"] - lam * er.float()).backward()
    model = build_model(fx).eval()
    eng = model.engine()
    eng.erank_weight = lam
    eng.use_paired_launches = paired
    d = [t.cuda() for t in batch]
    eng.forward(d[0], d[1], d[2], "train", training=False)
    eng.losses(d[2], d[4], d[3])
    eng.backward()
    torch.cuda.synchronize()
    assert abs(float(eng.erank_value()) - float(er)) < 5e-3 * max(1.0, float(er) / 50)
    for n, p in tr.p.items():
        if p.grad is not None:
            # (the fixture's fused tokens have sigma_min / sigma_max = 4e-5: in fp32 the singular vectors of the smallest
            #  singular values carry a relative error ~ eps * sigma_max / sigma_k ~ 3e-3, and d erank / d sigma is
            #  LARGEST there (log p_k -> -inf) -- the fp64 autograd checker cannot be met tighter than that)
            close_rel(eng.arena.g(n), p.grad, f"{tag}/erank-penalised grad {n}", rtol=1e-2)


@pytest.mark.parametrize("B,S,H,side", [(8, 32, 128, True), (8, 64, 512, True), (8, 64, 512, False), (8, 16, 1024, True),
                                        (16, 16, 1024, True)])
def test_effective_rank_penalty_baseline_shapes(B, S, H, side, oracle_lib):
    """The rank-regularised step at the per-GPU shapes of BASELINE configs[2..4] (cfg3: N=256,H=128 -- one CU's LDS; cfg4:
    [512,512] -- two-level block Jacobi; cfg5: [128|256, 1024] -- block Jacobi on fused^T): effective rank within the
    north-star's +-0.5 of svdvals (measured: <= 5e-3) and every gradient of  L - w * erank(fused)  against autograd
    through torch.linalg.svdvals (fp64) on the oracle.  No reference counterpart exists (the reference only
    describes the quantity, README.md:8-14); tolerance 1e-2 of each tensor's scale, see
    test_effective_rank_penalty_gradients for why fp32 singular vectors of the smallest sigma cannot do better."""
    from oracle import synth
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    lam, K = 0.05, 17
    pad = K + 1
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    model = FUTR(K, H, pad, torch.device("cuda"), args, n_query=8, n_head=8, num_encoder_layers=2, num_decoder_layers=1)
    names = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    params = {n: torch.from_numpy(v) for n, v in synth.fill_state(names).items()}
    model.load_state_dict(params, strict=False)
    model = model.to("cuda").eval()
    batch = [torch.from_numpy(x) for x in synth.make_batch(B, S, K, pad, 77)]
    tr = O.CpuTrainer(params, pad, 8, 1)
    out, aux = O.forward(tr.p, (batch[0], batch[2]), batch[1], "train", pad, 8, 1)
    res = O.losses(out, batch[2], batch[3], batch[4], pad)
    er = O.effective_rank_torch(aux["fused"].reshape(-1, H).double())
    (res["loss"] - lam * er.float()).backward()
    eng = model.engine()
    eng.erank_weight = lam
    eng.erank_side_stream = side
    d = [t.cuda() for t in batch]
    eng.forward(d[0], d[1], d[2], "train", training=False)
    eng.losses(d[2], d[4], d[3])
    eng.backward()
    torch.cuda.synchronize()
    w = eng.last["w"]
    from r3d_amd import ops
    assert (w.er_blk is None) == ops.erank_fits(B * S, H)
    got = float(eng.erank_value())
    assert abs(got - float(er)) < 5e-3 * max(1.0, float(er) / 50), (got, float(er))
    assert float(w.er_stats[0, 3]) < eng.erank_max_sweeps, "the sweeps enqueued must cover convergence"
    worst = (0.0, "")
    for n, p in tr.p.items():
        if p.grad is not None:
            g, r = eng.arena.g(n).double().cpu(), p.grad.double()
            rel = float((g - r).abs().max()) / max(float(r.abs().max()), 1e-5)
            worst = max(worst, (rel, n))
    print(f"[erank penalty B{B} S{S} H{H}] erank {got:.4f} vs svdvals {float(er):.4f}, sweeps {float(w.er_stats[0, 3]):.0f}, "
          f"worst gradient error / scale {worst[0]:.2e} ({worst[1]})")
    assert worst[0] <= 1e-2, worst


def test_effective_rank_warm_start(oracle_lib):
    """erank_warm_start: the Jacobi sweep of step t runs on X V0 with V0 from step t-1's backward.  Same singular
    values / effective rank (against svdvals on the CPU copy of the step's fused tokens), same gradients as the cold
    start at the same parameters, fewer sweeps once the basis is warm."""
    lam = 0.05
    fx = load_fixture("step_cfg2")
    m = fx["meta"]
    d = [t.cuda() for t in fixture_batch(fx)]
    model = build_model(fx).eval()
    eng = model.engine()
    eng.erank_weight = lam
    eng.erank_warm_start = True
    sweeps = []
    for it in range(4):
        eng.forward(d[0], d[1], d[2], "train", training=False)
        eng.losses(d[2], d[4], d[3], tick=True)
        eng.backward()
        torch.cuda.synchronize()
        w = eng.last["w"]
        sweeps.append(float(w.er_stats[0, 3]))
        want = O.effective_rank(w.fused.cpu().reshape(-1, m["H"]))
        assert abs(float(eng.erank_value()) - want) < 5e-3 * max(1.0, want / 50), (it, float(eng.erank_value()), want)
        vt = w.er_vt.double().cpu()
        assert float((vt @ vt.t() - torch.eye(vt.shape[0], dtype=torch.float64)).abs().max()) < 1e-4
        if it < 3:
            # a late-training-sized update (at lr 1e-3 this fixture's analytic parameters move the tokens by 20-100 %
            # per step, which no basis survives; the warm start then simply costs what a cold one does)
            eng.adamw(2e-6, 5e-3, ticked=True)
    assert max(sweeps[1:]) < sweeps[0], sweeps
    warm = {n: eng.arena.g(n).clone() for n in eng.arena.live_names}
    eng.erank_warm_start = False
    eng.forward(d[0], d[1], d[2], "train", training=False)
    eng.losses(d[2], d[4], d[3])
    eng.backward()
    torch.cuda.synchronize()
    assert float(eng.last["w"].er_stats[0, 3]) > max(sweeps[1:])
    for n, g in warm.items():
        close_rel(g, eng.arena.g(n).cpu(), f"warm vs cold erank-penalised grad {n}", rtol=3e-3)


def test_deferred_tail_one_launch_equals_three(oracle_lib):
    """engine.defer_tail: the decoder tail's forward, the losses and the tail's backward as ONE launch
    (r3d_decoder_tail_losses) give the same outputs, losses, counters and gradients as the three separate launches,
    with and without dropout (same Philox offset => same masks)."""
    import bench
    c = dict(bench.CFG)
    for training in (False, True):
        res = []
        for defer in (False, True):
            model = bench.build_model(c, torch.device("cuda"))
            if not training:
                model.eval()
            eng = model.engine()
            eng.defer_tail = defer
            feats, depth, lab, dur, tgt = bench.make_inputs(c, torch.device("cuda"), seed=5)
            out = eng.forward(feats, depth, lab, "train", training)
            loss, counts = eng.losses(lab, tgt, dur, tick=True)
            assert bool(getattr(eng.last["w"], "_tail_done", False)) == defer
            eng.backward()
            torch.cuda.synchronize()
            res.append(dict(loss=loss.clone(), counts=counts.clone(), grads=eng.arena.grads.clone(),
                            act=out["action"].clone(), dur=out["duration"].clone(), step=int(eng.step_t)))
        a, b = res
        # (not bitwise: the compiler contracts the two-term head dot products into differently ordered FMAs)
        assert torch.allclose(a["act"], b["act"], rtol=1e-5, atol=1e-6), float((a["act"] - b["act"]).abs().max())
        assert torch.allclose(a["dur"], b["dur"], rtol=1e-5, atol=1e-6), float((a["dur"] - b["dur"]).abs().max())
        assert torch.allclose(a["loss"], b["loss"], rtol=1e-5, atol=1e-7), (a["loss"], b["loss"])
        assert torch.equal(a["counts"], b["counts"]) and a["step"] == b["step"]
        d = (a["grads"] - b["grads"]).abs().max()
        assert float(d) <= 2e-5 * float(a["grads"].abs().max()), (float(d), float(a["grads"].abs().max()))


def test_deferred_loss_reduction_in_the_adamw_launch():
    """engine.defer_loss_reduce: the loss kernel leaves its per-unit partials unreduced (no arrival atomics, no
    last-workgroup pass) and one extra workgroup of the AdamW launch -- or r3d_losses_finalize on paths without that
    launch -- writes the same losses and counters; counters tick, gradients and the parameter update are bit-identical."""
    import bench
    c = dict(bench.CFG)
    for prefill in (True, False):
        res = []
        for defer in (False, True):
            model = bench.build_model(c, torch.device("cuda"))
            eng = model.engine()
            eng.defer_tail, eng.defer_loss_reduce = True, defer
            feats, depth, lab, dur, tgt = bench.make_inputs(c, torch.device("cuda"), seed=7)
            for _ in range(2):
                eng.forward(feats, depth, lab, "train", True)
                loss, counts = eng.losses(lab, tgt, dur, tick=True)
                assert (getattr(eng, "_loss_pending", None) is not None) == defer
                eng.backward()
                grads = eng.arena.grads.clone()
                eng.adamw(c["lr"], c["wd"], ticked=True, prefill_dropout=prefill)
                assert getattr(eng, "_loss_pending", None) is None
            torch.cuda.synchronize()
            res.append(dict(loss=loss.clone(), counts=counts.clone(), grads=grads, params=eng.arena.params.clone(),
                            step=int(eng.step_t), off=int(eng.drop_offset)))
        a, b = res
        assert torch.equal(a["loss"], b["loss"]), (a["loss"], b["loss"])
        assert torch.equal(a["counts"], b["counts"]) and a["step"] == b["step"] == 2 and a["off"] == b["off"]
        assert torch.equal(a["grads"], b["grads"]) and torch.equal(a["params"], b["params"])
        assert float(a["loss"][3]) > 0 and int(a["counts"][1]) > 0
    # the stand-alone entry point on the partials the last (deferred) step left, with the running sums it can feed
    from r3d_amd import ops
    w = eng.last["w"]
    loss2, cnt2 = torch.zeros(4, device="cuda"), torch.zeros(4, dtype=torch.int64, device="cuda")
    accl, accc = torch.ones(4, dtype=torch.float64, device="cuda"), torch.full((4,), 5, dtype=torch.int64, device="cuda")
    ops.losses_finalize(ops.loss_finalize_job(w.loss_ws, w.B, w.S, eng.Q, True, eng.dur_den, loss2, cnt2, accl, accc))
    torch.cuda.synchronize()
    assert torch.equal(loss2, b["loss"]) and torch.equal(cnt2, b["counts"])
    assert torch.equal(accl, 1.0 + loss2.double()) and torch.equal(accc, 5 + cnt2)
