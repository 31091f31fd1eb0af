"""Two ranks sharing one MI355X (gloo rendezvous, world_size 2): the pixel-sharded depth projection
(r3d_amd.parallel.PixelShardedDepth) must give the same training step as the replicated data-parallel step -- same
losses, same gradients (the sharded rank holds the SUM over ranks in its pixel columns, exactly what the all-reduce
leaves in the replicated run), same parameters after AdamW."""
import argparse
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from tests.helpers import load_fixture, fixture_params, fixture_batch  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(fx):
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    m = fx["meta"]
    args = argparse.Namespace(input_dim=2048, seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    model = FUTR(m["n_class"], m["H"], m["pad_idx"], torch.device("cuda"), args, n_query=m["n_query"], n_head=m["n_head"],
                 num_encoder_layers=2, num_decoder_layers=m["n_dec"])
    model.load_state_dict(fixture_params(fx), strict=False)
    return model.to("cuda").eval()


def _run(fx, rank, pixel_shard, steps, prefetch):
    from r3d_amd.parallel import DataParallelStep
    m = fx["meta"]
    model = _model(fx)
    eng = model.engine()
    dp = DataParallelStep(eng, pixel_shard=pixel_shard)
    dp.broadcast_parameters()
    batches = [[t.cuda() for t in fixture_batch(fx, seed=100 + 10 * s + rank)] for s in range(steps)]
    rec = []
    if prefetch and dp.tp is not None:
        dp.tp.prefetch(batches[0][1].reshape(m["B"] * m["S"], -1), slot=0)
    for s, (feats, depth, lab, dur, tgt) in enumerate(batches):
        dp.prepare_duration_denominator(dur, m["pad_idx"])
        eng.forward_begin(feats, depth, lab, "train", False)
        if prefetch and dp.tp is not None and s + 1 < steps:          # next batch's exchange under this step
            dp.tp.prefetch(batches[s + 1][1].reshape(m["B"] * m["S"], -1), slot=(s + 1) % 2)
        if dp.tp is not None:
            dp.tp.exchange_forward(eng._fw["w"])
        eng.forward_finish()
        loss, _ = eng.losses(lab, tgt, dur)
        eng.backward()
        dp.wait_grads()
        torch.cuda.synchronize()
        g = eng.arena.grads.clone()
        dep = eng.last["w"].dep.clone()
        eng.adamw(m["lr"], m["wd"], grad_scale=dp.grad_scale)
        rec.append(dict(loss=loss.clone(), grads=g, dep=dep))
    if dp.tp is not None:
        dp.tp.sync_full_weight()
    torch.cuda.synchronize()
    return eng, dp, rec


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        fx = load_fixture("step_tiny")
        lr = fx["meta"]["lr"]
        engA, dpA, recA = _run(fx, rank, False, 2, False)
        msgs = []
        for prefetch in (False, True):
            engB, dpB, recB = _run(fx, rank, True, 2, prefetch)
            a, b = engA.arena, engB.arena
            o, n, _ = a.offsets["depth_projection.weight"]
            H, P = engA.H, engA.P
            cols = slice(dpB.tp.p0, dpB.tp.p0 + dpB.tp.Pr)
            for s in range(2):
                ra, rb = recA[s], recB[s]
                tol = 1e-5 if s == 0 else 2e-2          # step 2 starts from parameters that differ by AdamW's step-1
                                                         # sign sensitivity where |g| ~ 0 (see test_engine_gpu)
                assert torch.allclose(ra["loss"], rb["loss"], rtol=tol, atol=1e-6), (s, ra["loss"], rb["loss"])
                sc = float(ra["dep"].abs().max())
                assert float((ra["dep"] - rb["dep"]).abs().max()) <= tol * sc
                if s == 0:
                    ga, gb = ra["grads"], rb["grads"]
                    small = slice(0, a.bucket_small[1])
                    sc = float(ga[small].abs().max())
                    assert float((ga[small] - gb[small]).abs().max()) <= 1e-5 * sc
                    gwa, gwb = ga[o:o + n].view(H, P)[:, cols], gb[o:o + n].view(H, P)[:, cols]
                    sc = float(gwa.abs().max())
                    assert float((gwa - gwb).abs().max()) <= 1e-5 * sc, "sharded depth weight gradient"
            pa, pb = a.params[:a.n_live], b.params[:b.n_live]
            d = (pa - pb).abs()
            assert float(d.max()) <= 2 * 2.1 * lr, float(d.max())
            frac = float((d <= 1e-5 * (1 + pa.abs())).double().mean())
            assert frac > 0.97, frac
            # every rank ends with the same complete weight
            t = pb.clone()
            dist.broadcast(t, src=0)
            assert torch.equal(t, pb)
            msgs.append((prefetch, frac))
        q.put((rank, "ok", msgs))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_pixel_sharded_depth_matches_replicated_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"


def _train_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        import tempfile
        from r3d_amd.train_proposed_depth import train
        from r3d_amd.optim import FlatAdamW
        fx = load_fixture("step_tiny")
        m = fx["meta"]
        batches = [fixture_batch(fx, seed=300 + 10 * s + rank) for s in range(4)]
        val = [[t[:1] for t in fixture_batch(fx, seed=999)]]
        finals = []
        for graph_steps in (False, True):
            model = _model(fx)
            args = argparse.Namespace(epochs=1, input_type="i3d_transcript", seg=True, anticipate=True, task="long",
                                      min_batch=1, graph_steps=graph_steps)

            class NoSched:
                def step(self):
                    pass
            opt = FlatAdamW(model.parameters(), 1e-3, weight_decay=5e-3)
            with tempfile.TemporaryDirectory() as d:
                train(args, model, batches, opt, NoSched(), None, d, m["pad_idx"], torch.device("cuda"), val, seed=rank)
            torch.cuda.synchronize()
            finals.append(model.engine().arena.params.clone())
        assert torch.equal(finals[0], finals[1]), float((finals[0] - finals[1]).abs().max())
        t = finals[1].clone()
        dist.broadcast(t, src=0)
        assert torch.equal(t, finals[1])                       # the ranks stayed in lock-step
        q.put((rank, "ok", ""))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_train_loop_graphed_data_parallel_equals_eager_two_ranks():
    """train() under replicated data parallelism (2 ranks, gloo): replaying each step as three hipGraphs around the two
    gradient all-reduces gives bit-identical parameters to enqueueing it launch by launch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"


class _DistAsRccl:
    """Test double with r3d_amd.rccl.RcclComm's interface over a torch.distributed group (gloo): rehearses RcclStep's
    exchange logic with two ranks on one GPU, where RCCL itself cannot form a communicator."""

    def __init__(self):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def group(self):
        import contextlib
        return contextlib.nullcontext()

    def all_reduce(self, t, stream=None):
        dist.all_reduce(t)

    def reduce_scatter_inplace(self, full, stream=None):
        n = full.numel() // self.world
        tmp = full.clone()
        dist.all_reduce(tmp)
        mine = full.view(-1)[self.rank * n:(self.rank + 1) * n]
        mine.copy_(tmp.view(-1)[self.rank * n:(self.rank + 1) * n])       # the other blocks stay partial, as with RCCL
        return mine

    def all_gather(self, out, inp, stream=None):
        parts = [torch.empty_like(inp) for _ in range(self.world)]
        dist.all_gather(parts, inp)
        out.view(self.world, -1).copy_(torch.stack([p.reshape(-1) for p in parts]))

    def all_to_all(self, recv, send, stream=None):
        parts = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(parts, send)
        blk = send.numel() // self.world
        for j in range(self.world):
            recv.view(-1)[j * blk:(j + 1) * blk].copy_(parts[j].view(-1)[self.rank * blk:(self.rank + 1) * blk])


def _run_rccl_step(fx, rank, pixel_shard, steps, blocked=False, fuse_adam=False):
    from r3d_amd.parallel import DataParallelStep, RcclStep
    m = fx["meta"]
    model = _model(fx)
    eng = model.engine()
    dp = DataParallelStep(eng, pixel_shard=pixel_shard)
    dp.broadcast_parameters()
    rs = RcclStep(dp, _DistAsRccl(), _DistAsRccl(), m["lr"], m["wd"], fuse_adam=fuse_adam)
    batches = [[t.cuda() for t in fixture_batch(fx, seed=100 + 10 * s + rank)] for s in range(steps)]
    x2d = [b[1].reshape(m["B"] * m["S"], -1) for b in batches]
    if blocked:                              # resident input already pixel-block-major [W, N, P/W]
        W = dist.get_world_size()
        x2d = [x.view(x.shape[0], W, -1).transpose(0, 1).contiguous() for x in x2d]
    rs.stage(x2d[0], batches[0][3], m["pad_idx"], 0)
    losses = []
    for s, (feats, depth, lab, dur, tgt) in enumerate(batches):
        if s + 1 < steps:                                             # the next step's inputs, one step ahead
            rs.stage(x2d[s + 1], batches[s + 1][3], m["pad_idx"], (s + 1) % 2)
        rs.run(feats, depth, lab, dur, tgt, m["pad_idx"], False, slot=s % 2)
        torch.cuda.synchronize()
        losses.append(eng.last["w"].loss.clone())
    if dp.tp is not None:
        dp.tp.sync_full_weight()
    torch.cuda.synchronize()
    return eng, losses


def _rccl_step_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        fx = load_fixture("step_tiny")
        lr = fx["meta"]["lr"]
        engA, _, recA = _run(fx, rank, False, 3, False)                # torch.distributed, replicated: the yardstick
        # (sharded?, resident input already pixel-block-major?, AdamW of the owned columns inside the GEMM epilogue?)
        for pixel_shard, blocked, fuse_adam in ((False, False, False), (True, False, False), (True, True, True)):
            engB, lossB = _run_rccl_step(fx, rank, pixel_shard, 3, blocked, fuse_adam)
            for s in range(3):
                tol = 1e-5 if s == 0 else 2e-2
                assert torch.allclose(recA[s]["loss"], lossB[s], rtol=tol, atol=1e-6), (pixel_shard, s, recA[s]["loss"],
                                                                                       lossB[s])
            a, b = engA.arena, engB.arena
            pa, pb = a.params[:a.n_live], b.params[:b.n_live]
            d = (pa - pb).abs()
            assert float(d.max()) <= 3 * 2.1 * lr, float(d.max())
            frac = float((d <= 1e-5 * (1 + pa.abs())).double().mean())
            assert frac > 0.95, (pixel_shard, frac)
            t = pb.clone()
            dist.broadcast(t, src=0)
            assert torch.equal(t, pb)
        q.put((rank, "ok", ""))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_rccl_step_flow_matches_replicated(world):
    """RcclStep (exchanges in stream order: reduce-scatter of the partial products, all-gather of d(depth_pre), the
    staged all-to-all and denominator, grouped bucket all-reduce) with a gloo stand-in for the communicator == the
    torch.distributed replicated step, over three steps with different batches (so a wrong slot or a stale prefetch
    shows); 2 and 4 ranks sharing the GPU."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"


def _train_rccl_worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        torch.cuda.set_device(0)
        import tempfile
        from r3d_amd.train_proposed_depth import train
        from r3d_amd.optim import FlatAdamW
        fx = load_fixture("step_tiny")
        m = fx["meta"]
        batches = [fixture_batch(fx, seed=300 + 10 * s) for s in range(4)]
        val = [[t[:1] for t in fixture_batch(fx, seed=999)]]
        finals = []
        for rehearse in (False, True):
            if rehearse:                     # one-rank RCCL group: train() takes the data-parallel RcclStep path
                dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
                os.environ["R3D_REHEARSE_DIST"] = "1"
            model = _model(fx)
            args = argparse.Namespace(epochs=1, input_type="i3d_transcript", seg=True, anticipate=True, task="long",
                                      min_batch=1, graph_steps=True)

            class NoSched:
                def step(self):
                    pass
            opt = FlatAdamW(model.parameters(), 1e-3, weight_decay=5e-3)
            import contextlib
            import io
            out = io.StringIO()
            with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(out):
                train(args, model, batches, opt, NoSched(), None, d, m["pad_idx"], torch.device("cuda"), val, seed=0)
            torch.cuda.synchronize()
            assert ("RCCL on the launch stream" in out.getvalue()) == rehearse, out.getvalue()
            finals.append(model.engine().arena.params.clone())
        os.environ.pop("R3D_REHEARSE_DIST", None)
        d = float((finals[0] - finals[1]).abs().max())
        assert d <= 1e-6, d
        q.put((0, "ok", d))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((0, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_train_loop_rccl_step_one_rank_equals_single_gpu():
    """train() on its data-parallel RcclStep path (RCCL calls captured inside the step's one hipGraph), rehearsed with a
    one-rank RCCL communicator, ends with the parameters of the plain one-GPU loop."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_train_rccl_worker, args=(_free_port(), q))
    p.start()
    rank, status, info = q.get(timeout=600)
    p.join(timeout=60)
    assert status == "ok", info


def _allreduce_flat_worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        from r3d_amd import _lib
        from r3d_amd.rccl import RcclComm
        lib = _lib.load()
        comm = RcclComm()
        x = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.25
        ref = x.clone()
        st = torch.cuda.current_stream().cuda_stream
        assert lib.r3d_allreduce_flat(x.data_ptr(), x.numel(), comm._comm, st) == 0
        assert lib.r3d_allreduce_flat(None, 4, comm._comm, st) == -1            # R3D_EINVAL
        assert lib.r3d_allreduce_flat(x.data_ptr(), 4, None, st) == -1
        assert lib.r3d_allreduce_flat(x.data_ptr(), 0, comm._comm, st) == 0
        comm.all_reduce(x)                      # the float32 path of RcclComm goes through the same entry point
        g = torch.cuda.CUDAGraph()              # ... and is capturable into a hipGraph like the step's other launches
        with torch.cuda.graph(g):
            x.mul_(2.0)
            comm.all_reduce(x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(x, ref * 2.0), float((x - ref * 2.0).abs().max())    # (capture records, the replay runs it once)
        comm.close()
        q.put((0, "ok", ""))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((0, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_allreduce_flat_c_abi_one_rank():
    """r3d_allreduce_flat (the C-ABI gradient exchange, include/r3d_hip.h) finds the RCCL copy PyTorch loaded, sums in
    place over a one-rank communicator (identity), rejects null arguments and replays from a hipGraph."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_allreduce_flat_worker, args=(_free_port(), q))
    p.start()
    rank, status, info = q.get(timeout=300)
    p.join(timeout=60)
    assert status == "ok", info


# ----------------------------------------------------------------------------------------------------------
# train(--pixel_shard) on the sharded RcclStep with a one-batch look-ahead
# ----------------------------------------------------------------------------------------------------------
def _train_sharded_worker(rank, world, port, q):
    """2 ranks sharing the GPU, gloo: train() with pixel_shard through _ShardedSteps (look-ahead staging, two slots; the
    RCCL binding replaced by the torch.distributed stand-in, so steps are enqueued, not captured) must end with the
    parameters of the launch-by-launch sharded loop (graph_steps=False) on the same batches."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        import contextlib
        import io
        import tempfile
        from r3d_amd.train_proposed_depth import train
        from r3d_amd.optim import FlatAdamW
        fx = load_fixture("step_tiny")
        m = fx["meta"]
        batches = [fixture_batch(fx, seed=500 + 10 * s + rank) for s in range(5)]
        batches.insert(2, None)                                   # the loader's None items are skipped (:128)
        val = [[t[:1] for t in fixture_batch(fx, seed=999)]]
        finals, logs = [], []
        for lookahead in (False, True):
            model = _model(fx)
            args = argparse.Namespace(epochs=2, input_type="i3d_transcript", seg=True, anticipate=True, task="long",
                                      min_batch=1, pixel_shard=True, graph_steps=lookahead, sharded_graphs=False,
                                      comm_factory=_DistAsRccl if lookahead else None)

            class NoSched:
                def step(self):
                    pass
            opt = FlatAdamW(model.parameters(), 1e-3, weight_decay=5e-3)
            out = io.StringIO()
            with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(out):
                train(args, model, batches, opt, NoSched(), None, d, m["pad_idx"], torch.device("cuda"), val, seed=0)
            torch.cuda.synchronize()
            finals.append(model.engine().arena.params.clone())
            logs.append(out.getvalue())
        assert "one hipGraph per step" in logs[1] or rank != 0
        a = model.engine().arena
        d = (finals[0][:a.n_live] - finals[1][:a.n_live]).abs()
        assert float(d.max()) <= 10 * 2.1e-3, float(d.max())       # (ill-conditioned AdamW elements move by +-lr per step)
        assert float((d <= 1e-5 * (1 + finals[0][:a.n_live].abs())).double().mean()) > 0.95
        ep = [ln for ln in logs[0].splitlines() if ln.startswith("Epoch")]
        eq = [ln for ln in logs[1].splitlines() if ln.startswith("Epoch")]
        assert ep == eq or rank != 0, (ep, eq)
        q.put((rank, "ok", ""))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_train_loop_pixel_shard_lookahead_equals_eager_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"


def _train_sharded_graph_worker(port, q, case="equal"):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        torch.cuda.set_device(0)
        import contextlib
        import io
        import tempfile
        from r3d_amd.train_proposed_depth import train
        from r3d_amd.optim import FlatAdamW
        fx = load_fixture("step_tiny")
        m = fx["meta"]
        batches = [fixture_batch(fx, seed=300 + 10 * s) for s in range(6)]
        epochs = 2
        if case == "ragged":                  # pad_sequence pads to the batch's longest clip: shapes alternate S, S - 1
            batches = [b if i % 2 == 0 else [b[0][:, :-1].contiguous(), b[1][:, :-1].contiguous(), b[2][:, :-1].contiguous(),
                                              b[3], b[4]] for i, b in enumerate(batches)]
        elif case == "short_epochs":          # two batches per epoch: the first batch of every epoch is never pre-staged
            batches, epochs = batches[:2], 4
        val = [[t[:1] for t in fixture_batch(fx, seed=999)]]
        finals = []
        for rehearse in (False, True):
            if rehearse:                     # one-rank RCCL group: train(--pixel_shard) takes the graphed look-ahead path
                dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
                os.environ["R3D_REHEARSE_DIST"] = "1"
            model = _model(fx)
            args = argparse.Namespace(epochs=epochs, input_type="i3d_transcript", seg=True, anticipate=True, task="long",
                                      min_batch=1, graph_steps=True, pixel_shard=rehearse)

            class NoSched:
                def step(self):
                    pass
            opt = FlatAdamW(model.parameters(), 1e-3, weight_decay=5e-3)
            out = io.StringIO()
            with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(out):
                train(args, model, batches, opt, NoSched(), None, d, m["pad_idx"], torch.device("cuda"), val, seed=0)
            torch.cuda.synchronize()
            finals.append(model.engine().arena.params.clone())
        os.environ.pop("R3D_REHEARSE_DIST", None)
        # (the sharded projection takes other GEMM routes -- other roundings -- than the one-GPU step: elements whose gradient
        #  is rounding noise move by +-lr per step in either, so the criterion is the one of the other sharded tests)
        a = model.engine().arena
        dd = (finals[0][:a.n_live] - finals[1][:a.n_live]).abs()
        assert float(dd.max()) <= 12 * 2.1e-3, float(dd.max())
        frac = float((dd <= 1e-5 * (1 + finals[0][:a.n_live].abs())).double().mean())
        assert frac > 0.95, frac
        q.put((0, "ok", frac))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((0, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("case", ["equal", "ragged", "short_epochs"])
def test_train_loop_pixel_shard_graphed_one_rank_equals_single_gpu(case):
    """train(--pixel_shard) on the sharded one-graph RcclStep (step graphs + staging graphs, two slots, look-ahead),
    rehearsed with a one-rank RCCL communicator, ends with the parameters of the plain one-GPU loop -- with equal batch
    shapes, with shapes that alternate (a ragged loader: un-staged batches whose staging is captured on the spot) and with
    two-batch epochs (the first batch of every epoch is never pre-staged)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_train_sharded_graph_worker, args=(_free_port(), q, case))
    p.start()
    rank, status, info = q.get(timeout=600)
    p.join(timeout=60)
    assert status == "ok", info


# ---------------------------------------------------------------------------------------------------------------------
# two REAL devices, real RCCL (skipped, loudly, on a one-GPU box): first contact with ncclAllToAll / ReduceScatter / the
# grouped AllGather + AllReduce between devices, eager and captured into the step's hipGraph
# ---------------------------------------------------------------------------------------------------------------------
def _two_device_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        from r3d_amd.parallel import DataParallelStep, RcclStep
        from r3d_amd.rccl import RcclComm
        fx = load_fixture("step_tiny")
        m = fx["meta"]
        steps = 3
        batches = [[t.cuda() for t in fixture_batch(fx, seed=100 + 10 * s + rank)] for s in range(steps)]
        x2d = [b[1].reshape(m["B"] * m["S"], -1) for b in batches]
        # yardstick: the torch.distributed (ProcessGroupNCCL) replicated step, launch by launch
        engA, _, recA = _run(fx, rank, False, steps, False)
        info = []
        for pixel_shard in (False, True):
            for graphed in (False, True):
                model = _model(fx)
                eng = model.engine()
                dp = DataParallelStep(eng, pixel_shard=pixel_shard)
                dp.broadcast_parameters()
                comm, side = RcclComm(), RcclComm()
                assert comm.nranks() == world and side.nranks() == world
                rs = RcclStep(dp, comm, side, m["lr"], m["wd"], fuse_adam=False)
                losses = []
                rs.stage(x2d[0], batches[0][3], m["pad_idx"], 0)
                graphs = {}
                for s, (feats, depth, lab, dur, tgt) in enumerate(batches):
                    if s + 1 < steps:
                        rs.stage(x2d[s + 1], batches[s + 1][3], m["pad_idx"], (s + 1) % 2)
                    if graphed and s >= 1:
                        # capture this slot's step (the RCCL kernels become graph nodes) and replay it: the static buffers
                        # are the batch tensors of step s, so the replay IS step s
                        torch.cuda.synchronize()
                        g = torch.cuda.CUDAGraph()
                        eng._drop_ready = None
                        with torch.cuda.graph(g):
                            rs.run(feats, depth, lab, dur, tgt, m["pad_idx"], False, slot=s % 2)
                        graphs[s] = g
                        # (capture does not execute: the replay runs the step once)
                        g.replay()
                    else:
                        rs.run(feats, depth, lab, dur, tgt, m["pad_idx"], False, slot=s % 2)
                    torch.cuda.synchronize()
                    losses.append(eng.last["w"].loss.clone())
                if dp.tp is not None:
                    dp.tp.sync_full_weight()
                torch.cuda.synchronize()
                for s in range(steps):
                    tol = 1e-5 if s == 0 else 2e-2
                    assert torch.allclose(recA[s]["loss"], losses[s], rtol=tol, atol=1e-6), (pixel_shard, graphed, s,
                                                                                           recA[s]["loss"], losses[s])
                a, b = engA.arena, eng.arena
                pa, pb = a.params[:a.n_live], b.params[:b.n_live]
                d = (pa - pb).abs()
                assert float(d.max()) <= steps * 2.1 * m["lr"], float(d.max())
                frac = float((d <= 1e-5 * (1 + pa.abs())).double().mean())
                assert frac > 0.95, (pixel_shard, graphed, frac)
                t = pb.clone()
                dist.broadcast(t, src=0)
                assert torch.equal(t, pb), "ranks must end with identical parameters"
                info.append((pixel_shard, graphed, frac))
                comm.close()
                side.close()
        q.put((rank, "ok", info))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_step_on_two_real_devices():
    """Replicated AND pixel-sharded RcclStep, eager AND with the step captured into a hipGraph, on two real devices over
    real RCCL (xGMI), against the torch.distributed replicated step.  A one-GPU box cannot run it: the skip says so --
    the first multi-GPU node that runs the suite executes the exchanges between devices before any benchmark does
    (reference counterpart: nn.DataParallel, main_darai.py:133)."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"NOT RUN: needs >= 2 MI355X devices, this box has {n} -- RCCL between devices is unexercised here")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_device_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"
