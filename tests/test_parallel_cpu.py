"""The data-parallel host logic (r3d_amd/parallel.py) on CPU with the gloo backend, world_size 2: bucketed gradient
all-reduce + 1/world scale, the global duration denominator and the eval-mode score all-reduce must reproduce the
single-process (global batch) result of the oracle.  The oracle stands in for the HIP compute here (tests only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import futr_oracle as O, synth
from tests.helpers import load_fixture, fixture_params


class _FakeArena:
    def __init__(self, names, params):
        self.names = names
        sizes = [params[n].numel() for n in names]
        self.offs = np.concatenate([[0], np.cumsum(sizes)])
        self.grads = torch.zeros(int(self.offs[-1]))
        cut = int(self.offs[names.index("depth_projection.weight")])
        # same bucket layout idea as ParamArena: the big weight gradient is its own bucket
        self.order = [n for n in names if n != "depth_projection.weight"] + ["depth_projection.weight"]
        sizes = [params[n].numel() for n in self.order]
        self.offs = np.concatenate([[0], np.cumsum(sizes)])
        self.bucket_small = (0, int(self.offs[-2]))
        self.bucket_big = (int(self.offs[-2]), int(self.offs[-1]))

    def view(self, n):
        i = self.order.index(n)
        return self.grads[int(self.offs[i]):int(self.offs[i + 1])]


class _FakeEngine:
    def __init__(self, arena):
        self.arena = arena
        self.grad_hook = None
        self.score_allreduce = None
        self.dur_den = None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from r3d_amd.parallel import DataParallelStep, shard_range
    fx = load_fixture("train_loop")                      # H=64, B=8, S=6
    m = fx["meta"]
    params = fixture_params(fx)
    full = [torch.from_numpy(x) for x in synth.make_batch(m["B"], m["S"], m["n_class"], m["pad_idx"], m["seed"])]
    lo, hi = shard_range(m["B"], rank, world)
    shard = [t[lo:hi] for t in full]
    live = [n for n in fx["param_names"] if O.is_live(n)]
    eng = _FakeEngine(_FakeArena(live, params))
    dp = DataParallelStep(eng)
    assert dp.world == world and eng.dur_den is not None
    # ---- one data-parallel step on the shard
    dp.prepare_duration_denominator(shard[3], m["pad_idx"])
    tr = O.CpuTrainer(params, m["pad_idx"], m["n_head"], 1)
    for p_ in tr.p.values():
        p_.grad = None
    out, _ = O.forward(tr.p, (shard[0], shard[2]), shard[1], "train", m["pad_idx"], m["n_head"], 1)
    res = O.losses(out, shard[2], shard[3], shard[4], m["pad_idx"], dur_den=float(eng.dur_den))
    res["loss"].backward()
    for n in live:
        eng.arena.view(n).copy_(tr.p[n].grad.reshape(-1))
    eng.grad_hook("small_ready")
    eng.grad_hook("big_ready")
    dp.wait_grads()
    eng.arena.grads.mul_(dp.grad_scale)
    # ---- eval-mode score all-reduce
    sums = torch.stack([shard[0].reshape(-1, 2048)[:, :64].abs().double().sum(0),
                        shard[0].reshape(-1, 2048)[:, 64:128].abs().double().sum(0)])
    count = eng.score_allreduce(sums, shard[0].shape[0] * shard[0].shape[1])
    if rank == 0:
        ref = O.CpuTrainer(params, m["pad_idx"], m["n_head"], 1)
        rres, _, _ = ref.step(full, apply=False)
        worst = 0.0
        for n in live:
            g, r = eng.arena.view(n), ref.p[n].grad.reshape(-1)
            # parameters whose true gradient is 0 (e.g. fc_len.bias) hold rounding noise only: absolute floor
            worst = max(worst, float((g - r).abs().max()) / max(float(r.abs().max()), 1e-4))
        gsum = torch.stack([full[0].reshape(-1, 2048)[:, :64].abs().double().sum(0),
                            full[0].reshape(-1, 2048)[:, 64:128].abs().double().sum(0)])
        q.put(dict(worst=worst, den=float(eng.dur_den), den_ref=float((full[3] != m["pad_idx"]).sum()) / world,
                   score_err=float((sums - gsum).abs().max()), count=count, n_rows=m["B"] * m["S"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_step_equals_global_batch(oracle_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["worst"] < 1e-4, res            # averaged shard gradients == gradient of the global-batch loss
    assert abs(res["den"] - res["den_ref"]) < 1e-6
    assert res["score_err"] < 1e-9 and res["count"] == res["n_rows"]


def test_shard_range_partitions_everything():
    from r3d_amd.parallel import shard_range
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1


# ---- pixel-sharded depth projection: the exchange logic with ragged per-rank batches (CPU stand-in for the GEMM) ----
class _TpArena:
    def __init__(self, H, P, seed):
        g = torch.Generator().manual_seed(seed)
        self.offsets = {"depth_projection.weight": (8, H * P, (H, P))}
        self.params = torch.randn(8 + H * P, generator=g)
        self.grads = torch.zeros(8 + H * P)
        self.exp_avg = torch.zeros(8 + H * P)
        self.exp_avg_sq = torch.zeros(8 + H * P)

    def p(self, n):
        o, k, shp = self.offsets[n]
        return self.params[o:o + k].view(shp)


class _TpEngine:
    def __init__(self, H, P):
        self.H, self.P, self.device, self.tp = H, P, torch.device("cpu"), None
        self.arena = _TpArena(H, P, 7)


def _cpu_gemm(layout, A, B, C, ws=None):
    from r3d_amd._lib import GEMM_NT
    C.copy_(A @ B.t() if layout == GEMM_NT else A.t() @ B)


class _W:
    pass


def _tp_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        from r3d_amd.parallel import PixelShardedDepth
        H, P = 8, 48
        rows = [5, 3]                                           # ragged: the last batch of an epoch
        g = torch.Generator().manual_seed(3)
        X = [torch.randn(r, P, generator=g) for r in rows]       # every rank can rebuild every rank's inputs
        D = [torch.randn(r, H, generator=g) for r in rows]
        eng = _TpEngine(H, P)
        tp = PixelShardedDepth(eng, equal_batches=False)
        tp.gemm = _cpu_gemm
        Wfull = eng.arena.p("depth_projection.weight").clone()
        w = _W()
        for use_prefetch in (False, True):
            if use_prefetch:
                tp.prefetch(X[rank], slot=1)
            tp.partial_forward(w, X[rank], None)
            tp.exchange_forward(w)
            got = tp.summed(w)
            assert torch.allclose(got, X[rank] @ Wfull.t(), atol=1e-5), "forward: sum of the ranks' pixel blocks"
            w.d_dep_pre = D[rank]
            tp.exchange_backward(w)
            tp.wgrad(w, None)
            want = sum(d.t() @ x for d, x in zip(D, X))          # what the all-reduce of the replicated run would hold
            assert torch.allclose(tp.g, want[:, tp.p0:tp.p0 + tp.Pr], atol=1e-5), "weight gradient of the owned columns"
        # each rank updates only its columns; sync_full_weight() restores one complete, identical weight everywhere
        tp.w.add_(float(rank + 1))
        tp.sync_full_weight()
        full = eng.arena.p("depth_projection.weight")
        for r in range(world):
            blk = slice(r * tp.Pr, (r + 1) * tp.Pr)
            assert torch.allclose(full[:, blk], Wfull[:, blk] + float(r + 1))
        q.put((rank, "ok", ""))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_pixel_sharded_depth_exchange_two_ranks_ragged():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"
