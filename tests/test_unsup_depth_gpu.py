"""GPU parity of the depth-as-query model (reference model/futr_unsupervised_depth.py) through the C ABI: outputs, the
three losses, counters, every gradient and one fused AdamW step against the oracle on the same seeded inputs and against
the fixtures generated from the imported reference (tests/golden/make_golden_unsup.py).  Train mode only: the reference's
validate() crashes on this model -- a tuple in mode='val' is accepted here and checked against the oracle.
Tolerance: 1e-3 relative (fp32)."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import futr_oracle as O  # noqa: E402
from tests.helpers import load_fixture, fixture_params, stats, assert_close  # noqa: E402
from tests.test_unsup_depth_cpu import unsup_batch  # noqa: E402
from tests.test_engine_gpu import close_rel  # noqa: E402

RTOL = 1e-3


def build_model(fx):
    from r3d_amd.model.futr_unsupervised_depth import FUTR
    m = fx["meta"]
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    model = FUTR(m["n_class"], m["H"], m["pad_idx"], torch.device("cuda"), args, n_query=m["n_query"], n_head=m["n_head"],
                 num_encoder_layers=2, num_decoder_layers=m["n_dec"])
    missing = model.load_state_dict(fixture_params(fx), strict=False)
    assert not missing.unexpected_keys and all("pos_table" in k for k in missing.missing_keys)
    return model.to("cuda")


@pytest.mark.parametrize("tag", ["unsup_tiny", "unsup_h128", "unsup_dec2"])
def test_step_parity(tag, oracle_lib):
    fx = load_fixture(tag)
    m = fx["meta"]
    batch = unsup_batch(fx)
    tr = O.CpuTrainer(fixture_params(fx), m["pad_idx"], m["n_head"], m["n_dec"], m["lr"], m["wd"], unsup_depth=True,
                      n_query=m["n_query"])
    ores, oout, oaux = tr.step(batch, apply=False)
    model = build_model(fx).eval()
    eng = model.engine()
    d = [t.cuda() for t in batch]
    out = eng.forward(d[0], d[1], d[2], "train", training=False)
    torch.cuda.synchronize()
    w = eng.last["w"]
    close_rel(w.mem.view(m["B"], m["S"], m["H"]), oaux["memory"].detach(), f"{tag}/memory")
    close_rel(w.qpos.view(m["B"], m["S"], m["H"]), oaux["query"].detach(), f"{tag}/query")
    close_rel(w.tgtF.view(m["B"], m["S"], m["H"]), oaux["tgt"].detach(), f"{tag}/decoder output")
    close_rel(w.pooled.view(m["B"], m["n_query"], m["H"]), oaux["pooled"].detach(), f"{tag}/pooled")
    for k, fk in (("action", "out_action"), ("duration", "out_duration"), ("seg", "out_seg")):
        close_rel(out[k], oout[k].detach(), f"{tag}/{k} vs oracle")
        close_rel(out[k], fx[fk], f"{tag}/{k} vs reference fixture")
    loss, counts = eng.losses(d[2], d[4], d[3])
    eng.backward()
    torch.cuda.synchronize()
    assert_close(loss.cpu(), fx["losses"], RTOL, 1e-6, f"{tag}/losses vs fixture")
    assert counts.cpu().tolist() == fx["counts"].tolist()
    live = fx["live_names"]
    assert sorted(live) == sorted(n for n in fx["param_names"] if eng.arena.is_live(n))
    for n in live:
        close_rel(eng.arena.g(n), tr.p[n].grad, f"{tag}/grad {n}", rtol=2e-3)
    gs = np.stack([stats(eng.arena.g(n)) for n in live])
    ref = fx["grad_stats"]
    assert bool((np.abs(gs[:, 0] - ref[:, 0]) <= 2e-3 * ref[:, 0] + 1e-7).all()), "grad norms vs reference fixture"
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
    ps = np.stack([stats(eng.arena.p(n)) for n in live])
    # (parameters whose gradient is rounding noise -- e.g. the key bias of a softmax attention, true gradient 0 -- move
    #  by +-lr in any implementation: only the well-conditioned ones are compared with the reference's post-step norms)
    wellp = fx["post_well_frac"] > 0.99
    dn = np.abs(ps[:, 0] - fx["post_stats"][:, 0])
    assert bool((dn[wellp] <= 1e-3 * fx["post_stats"][wellp, 0] + 1e-6).all()), "post-AdamW norms"
    assert wellp.sum() >= len(live) // 2
    params = dict(model.named_parameters())
    for n in fx["param_names"]:
        if n not in live:
            assert torch.equal(params[n].cpu(), fixture_params(fx)[n]), n


def test_autograd_bridge_and_eval_mode(oracle_lib):
    fx = load_fixture("unsup_tiny")
    m = fx["meta"]
    batch = unsup_batch(fx)
    model = build_model(fx).eval()
    d = [t.cuda() for t in batch]
    out = model((d[0], d[2]), d[1])
    loss = (out["seg"] ** 2).mean() + out["action"].sum() * 0.01 + (out["duration"] * 0.1).exp().mean()
    loss.backward()
    torch.cuda.synchronize()
    tr = O.CpuTrainer(fixture_params(fx), m["pad_idx"], m["n_head"], m["n_dec"], unsup_depth=True)
    oout, _ = O.forward_unsup_depth(tr.p, (batch[0], batch[2]), batch[1], "train", m["pad_idx"], m["n_head"], m["n_dec"])
    ol = (oout["seg"] ** 2).mean() + oout["action"].sum() * 0.01 + (oout["duration"] * 0.1).exp().mean()
    ol.backward()
    for n, p in model.named_parameters():
        if tr.p[n].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            close_rel(p.grad, tr.p[n].grad, f"bridge grad {n}", rtol=2e-3)
    # any mode but 'train': no key-padding mask; the bare tensor (the reference's convention) and a tuple both work
    with torch.no_grad():
        o1 = model(d[0], d[1], mode="val")
        o2 = model((d[0], d[2]), d[1], mode="val")
        oo, _ = O.forward_unsup_depth(tr.p, batch[0], batch[1], "val", m["pad_idx"], m["n_head"], m["n_dec"])
    for k in ("action", "duration", "seg"):
        assert torch.equal(o1[k], o2[k])
        close_rel(o1[k], oo[k].detach(), f"val/{k}")


def test_dropout_and_train_loop(oracle_lib):
    """Training state: Philox dropout on both encodings and the decoder (statistics), the fused step decreases the loss,
    and r3d_amd.train_proposed_depth.train() drives this model end to end (epoch prints, checkpoint)."""
    import io
    import contextlib
    import tempfile
    import os
    from r3d_amd import train_proposed_depth as T
    from r3d_amd.optim import FlatAdamW
    fx = load_fixture("unsup_h128")
    m = fx["meta"]
    batch = unsup_batch(fx)
    model = build_model(fx).train()
    eng = model.engine()
    d = [t.cuda() for t in batch]
    eng.forward(d[0], d[1], d[2], "train", training=True)
    torch.cuda.synchronize()
    w = eng.last["w"]
    keep = float(w.drop["pe_rgb"].float().mean())
    assert abs(keep - 0.9) < 0.02
    assert abs(float((w.mem == 0).float().mean()) - 0.1) < 0.03
    losses = []
    for _ in range(6):
        loss, _ = eng.train_step(d[0], d[1], d[2], d[3], d[4], 1e-3, 5e-3, training=False)
        losses.append(float(loss[3]))
    assert losses[-1] < losses[0], losses
    args = argparse.Namespace(epochs=1, anticipate=True, seg=True, task="long", input_type="i3d_transcript")
    opt = FlatAdamW(model.parameters(), lr=1e-3, weight_decay=5e-3)

    class NoSched:
        def step(self):
            pass
    val = [[t[:1] for t in batch]]
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as dd, contextlib.redirect_stdout(buf):
        T.train(args, model, [batch, batch], opt, NoSched(), torch.nn.MSELoss(reduction="none"), dd, m["pad_idx"],
                torch.device("cuda"), val, 1)
        files = sorted(os.listdir(dd))
    log = buf.getvalue()
    assert "Epoch [ 1 / 1 ]" in log and "Validation Loss" in log
    assert files == [] or files == ["seed_1_best.ckpt", "seed_1_checkpoint0.ckpt"]
