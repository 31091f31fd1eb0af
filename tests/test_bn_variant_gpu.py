"""The BN-blend fuser variant (model/futr_safuser_batchnormalization.py) through the HIP engine against
the oracle and the fixtures generated from the imported reference: train-state step (batch statistics, running-stat
update, |gamma| selection, alpha blend), every gradient, and the eval-state forward on the updated running statistics."""
import argparse
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import futr_oracle as O, synth  # noqa: E402
from tests.helpers import load_fixture, fixture_params, assert_close  # noqa: E402
from tests.test_engine_gpu import close_rel  # noqa: E402


def _model(fx):
    from r3d_amd.model.futr_safuser_batchnormalization import FUTR
    m = fx["meta"]
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    hw = m["depth_hw"]
    model = FUTR(m["n_class"], m["H"], m["pad_idx"], torch.device("cuda"), args, n_query=8, n_head=8, num_encoder_layers=2,
                 num_decoder_layers=m["n_dec"], depth_pixels=hw[0] * hw[1])
    missing = model.load_state_dict(fixture_params(fx), strict=False)
    assert not missing.unexpected_keys
    assert all(("pos_table" in k) or ("running_" in k) or ("num_batches" in k) for k in missing.missing_keys), missing.missing_keys
    return model.to("cuda")


@pytest.mark.parametrize("tag", ["bn_tiny", "bn_cfg2"])
def test_bn_variant_step_parity(tag, oracle_lib):
    fx = load_fixture(tag)
    m = fx["meta"]
    batch = [torch.from_numpy(x) for x in synth.make_batch(m["B"], m["S"], m["n_class"], m["pad_idx"], m["seed"],
                                                           depth_hw=tuple(m["depth_hw"]))]
    C = m["H"]
    st0 = {}
    for pre in ("fuser.bn_rgb.", "fuser.bn_depth."):
        st0[pre + "running_mean"], st0[pre + "running_var"] = torch.zeros(C), torch.ones(C)
        st0[pre + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    tr = O.CpuTrainer(fixture_params(fx), m["pad_idx"], 8, m["n_dec"], bn_state=st0, bn_training=True)
    ores, oout, oaux = tr.step(batch, apply=False)
    model = _model(fx).eval()                         # dropout off; BatchNorm in its training state via bn_training
    eng = model.engine()
    assert eng.bn
    d = [t.cuda() for t in batch]
    out = eng.forward(d[0], d[1], d[2], "train", training=False, bn_training=True)
    torch.cuda.synchronize()
    for k, fk in (("action", "out_action"), ("duration", "out_duration"), ("seg", "out_seg")):
        close_rel(out[k], oout[k].detach(), f"{tag}/{k} vs oracle")
        close_rel(out[k], fx[fk], f"{tag}/{k} vs reference fixture")
    close_rel(eng.last["w"].fused.view(m["B"], m["S"], C), fx["fused"], f"{tag}/fused")
    assert np.array_equal(np.sort(eng.last["idx"][0].cpu().numpy()), fx["idx_rgb"])       # bit-exact selection
    assert np.array_equal(np.sort(eng.last["idx"][1].cpu().numpy()), fx["idx_dep"])
    loss, counts = eng.losses(d[2], d[4], d[3])
    eng.backward()
    torch.cuda.synchronize()
    assert_close(loss.cpu(), fx["losses"], 1e-3, 1e-6, f"{tag}/losses")
    assert counts.cpu().tolist() == fx["counts"].tolist()
    live = json.loads(str(fx["live_names"])) if not isinstance(fx["live_names"], list) else fx["live_names"]
    for n in live:
        close_rel(eng.arena.g(n), tr.p[n].grad, f"{tag}/grad {n}", rtol=2e-3)
    for n in ("fuser.alpha", "fuser.bn_rgb.weight", "fuser.bn_rgb.bias", "fuser.bn_depth.weight", "fuser.bn_depth.bias"):
        close_rel(eng.arena.g(n), fx["grad::" + n], f"{tag}/grad {n} vs reference fixture", rtol=2e-3)
    # running statistics after the step, then the eval-state forward on them
    sd = model.state_dict()
    for n in json.loads(str(fx["buffer_names"])):
        assert_close(sd[n].float().cpu(), fx["buf::" + n], 1e-4, 1e-6, n)
    out_e = eng.forward(d[0], d[1], d[2], "train", training=False, need_grad=False)
    torch.cuda.synchronize()
    for k, fk in (("action", "eval_action"), ("duration", "eval_duration"), ("seg", "eval_seg")):
        close_rel(out_e[k], fx[fk], f"{tag}/eval {k}")


def test_bn_variant_trains(oracle_lib):
    """A few fused training steps (dropout on) run and reduce the loss; parameters of the BatchNorms and alpha move."""
    fx = load_fixture("bn_cfg2")
    m = fx["meta"]
    batch = [torch.from_numpy(x).cuda() for x in synth.make_batch(m["B"], m["S"], m["n_class"], m["pad_idx"], m["seed"],
                                                                  depth_hw=tuple(m["depth_hw"]))]
    model = _model(fx).train()
    eng = model.engine()
    a0 = eng.arena.p("fuser.alpha").clone()
    first = None
    for i in range(8):
        loss, _ = eng.train_step(batch[0], batch[1], batch[2], batch[3], batch[4], 1e-3, 5e-3, training=True)
        if i == 0:
            first = float(loss[3])
    torch.cuda.synchronize()
    assert float(loss[3]) < first and torch.isfinite(loss).all()
    assert float((eng.arena.p("fuser.alpha") - a0).abs().max()) > 0
    assert int(model.fuser.bn_rgb.num_batches_tracked) == 8


def _sync_bn_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from r3d_amd.parallel import DataParallelStep
        fx = load_fixture("bn_tiny")
        m = fx["meta"]
        B, C = m["B"], m["H"]
        gb = [torch.from_numpy(x) for x in synth.make_batch(world * B, m["S"], m["n_class"], m["pad_idx"], m["seed"] + 3,
                                                            depth_hw=tuple(m["depth_hw"]))]
        st0 = {}
        for pre in ("fuser.bn_rgb.", "fuser.bn_depth."):
            st0[pre + "running_mean"], st0[pre + "running_var"] = torch.zeros(C), torch.ones(C)
            st0[pre + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        tr = O.CpuTrainer(fixture_params(fx), m["pad_idx"], 8, m["n_dec"], bn_state=st0, bn_training=True)
        tr.step(gb, apply=False)                          # the single-process reference view: one batch of world*B clips
        model = _model(fx).eval()
        eng = model.engine()
        dp = DataParallelStep(eng)
        assert eng.bn_sync is not None
        mine = [t[rank * B:(rank + 1) * B].cuda() for t in gb]
        dp.prepare_duration_denominator(mine[3], m["pad_idx"])
        eng.forward(mine[0], mine[1], mine[2], "train", training=False, bn_training=True)
        eng.losses(mine[2], mine[4], mine[3])
        eng.backward()
        dp.wait_grads()
        torch.cuda.synchronize()
        live = json.loads(str(fx["live_names"])) if not isinstance(fx["live_names"], list) else fx["live_names"]
        for n in live:
            close_rel(eng.arena.g(n) * dp.grad_scale, tr.p[n].grad, f"sync-bn grad {n}", rtol=2e-3)
        sd = model.state_dict()
        for pre in ("fuser.bn_rgb.", "fuser.bn_depth."):
            assert_close(sd[pre + "running_mean"].float().cpu(), tr.bn_state[pre + "running_mean"], 1e-4, 1e-6, pre + "mean")
            assert_close(sd[pre + "running_var"].float().cpu(), tr.bn_state[pre + "running_var"], 1e-4, 1e-6, pre + "var")
            assert int(sd[pre + "num_batches_tracked"]) == 1
        q.put((rank, "ok", ""))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_bn_variant_sync_statistics_two_ranks(oracle_lib):
    """Data parallel BN-blend variant (2 ranks on one GPU, gloo): with the global-batch BatchNorm statistics of
    parallel.SyncBatchNorm the averaged rank gradients and the running statistics equal the oracle's on the whole batch."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"
