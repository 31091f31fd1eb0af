"""Data-parallel training of the token-fusion step: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm) -- the MI355X replacement for the reference's single-process nn.DataParallel
(main_darai.py:133).  The reference has no collective of its own; the exchanges below are the three
cross-batch couplings of SURVEY.md 8(e):

 1. gradient all-reduce of the flat grad arena in two buckets: everything except depth_projection.weight goes out as
    soon as it is complete and travels while the [H, 50176] weight-gradient GEMM (the last and largest kernel of the
    backward) runs; the big bucket follows.  The 1/world average is folded into AdamW's grad_scale (no extra pass);
 2. the duration-loss denominator: the reference divides by the GLOBAL mask sum (train_proposed_depth.py:206-207), so
    each rank uses (global mask sum / world) -- one float all-reduced at step start;
 3. eval-mode selection scores: per-channel |x| sums (fp64) are all-reduced before the top-k so every rank selects the
    channels the single-process reference would (futr_safuser_tokenfusion.py:49-54).

The class only needs an object with the engine's hook attributes and tensors, so its logic is exercised on CPU with
the gloo backend in tests/test_parallel_cpu.py.

PixelShardedDepth (opt-in) removes the big bucket altogether: depth_projection.weight [H, 50176] -- 86 % of the
trainable parameters at H=128 -- becomes tensor-parallel over PIXELS.  Rank r owns columns [r*P/W, (r+1)*P/W) of the
weight, of its gradient and of both AdamW moments; what crosses xGMI instead of the 25.7 MB gradient is
  * the depth INPUT, all-to-all (each rank receives its pixel block of every rank's clips): half the bytes of the
    gradient all-reduce and, unlike it, known before the step starts -> prefetched under the previous step;
  * two [sum_r N_r, H] activations (512 KB at 8 GPUs): the partial products forward, d(depth_pre) backward.
The mathematics is unchanged (the sum over pixels is split across ranks instead of across split-K workgroups).
"""
import os

import torch
import torch.distributed as dist

from . import ops
from ._lib import GEMM_NT, GEMM_TN


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items clips for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PixelShardedDepth:
    """depth_projection (futr_safuser_tokenfusion.py:143,194-195) tensor-parallel over its 50176 input pixels."""

    def __init__(self, engine, process_group=None, equal_batches=True, input_group=None):
        """input_group: a second communicator for the input all-to-all, so that a prefetch does not queue ahead of the
        step's own (latency-critical) exchanges on the gradient communicator's stream."""
        self.eng, self.pg = engine, process_group
        self.pg_in = input_group if input_group is not None else process_group
        self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        P, H = engine.P, engine.H
        if P % (4 * self.world):
            raise ValueError(f"{P} depth pixels do not split into {self.world} 16-byte aligned column blocks")
        self.Pr = P // self.world
        self.p0 = self.rank * self.Pr
        a = engine.arena
        o, n, _ = a.offsets["depth_projection.weight"]
        cols = slice(self.p0, self.p0 + self.Pr)
        self.w = a.params[o:o + n].view(H, P)[:, cols]
        self.g = a.grads[o:o + n].view(H, P)[:, cols]
        self.m = a.exp_avg[o:o + n].view(H, P)[:, cols]
        self.v = a.exp_avg_sq[o:o + n].view(H, P)[:, cols]
        self.equal = equal_batches
        self.native_a2a = dist.get_backend(process_group) == "nccl"
        self.gemm = ops.gemm              # (tests substitute a CPU stand-in to rehearse the exchange logic)
        self.native_gather = self.native_a2a          # all_gather_into_tensor (RCCL); falls back to the all-reduce form
        self.in_stream = torch.cuda.Stream(engine.device) if engine.device.type == "cuda" else None
        self.bufs = {}
        self.ready = {}                   # data_ptr of a depth batch -> its exchanged shard (prefetched)
        engine.tp = self

    def _buf(self, key, shape):
        if key not in self.bufs:
            self.bufs[key] = torch.empty(*shape, dtype=torch.float32, device=self.eng.device)
        return self.bufs[key]

    def _rows(self, N):
        if self.equal:
            return [N] * self.world
        t = torch.zeros(self.world, dtype=torch.int64, device=self.eng.device)
        t[self.rank] = N
        dist.all_reduce(t, group=self.pg)
        return [int(x) for x in t.cpu()]

    # -- inputs ------------------------------------------------------------------------------------------------
    def shard_inputs(self, x_dep, slot=0, async_op=False):
        """[N, P] depth rows of THIS rank's clips -> [sum_r N_r, P/W]: this rank's pixel block of EVERY rank's clips
        (rows ordered by rank).  async_op: the exchange runs on the communicator's stream; inputs() joins it."""
        W, Pr = self.world, self.Pr
        N = x_dep.shape[0]
        rows = self._rows(N)
        tot = sum(rows)
        recv = self._buf(("recv", tot, slot), (tot, Pr))
        work = None
        if self.native_a2a:
            send = self._buf(("send", N, slot), (W, N, Pr))
            # a prefetch (async_op) does its 25.7 MB re-layout copy on a side stream too: nothing of it sits on the step's
            # critical path
            side = self.in_stream if (async_op and self.in_stream is not None) else None
            cur = torch.cuda.current_stream() if side is not None else None
            if side is not None:
                side.wait_stream(cur)
            with torch.cuda.stream(side) if side is not None else _null():
                send.copy_(x_dep.view(N, W, Pr).transpose(0, 1))
                if all(r == N for r in rows):
                    work = dist.all_to_all_single(recv, send.view(W * N, Pr), group=self.pg_in, async_op=async_op)
                else:
                    work = dist.all_to_all_single(recv, send.view(W * N, Pr), output_split_sizes=rows,
                                                  input_split_sizes=[N] * W, group=self.pg_in, async_op=async_op)
        else:                             # backends without all-to-all (gloo rehearsal): W broadcasts
            off = 0
            for j in range(W):
                full = x_dep if j == self.rank else self._buf(("bc", rows[j]), (rows[j], self.eng.P))
                dist.broadcast(full, src=dist.get_global_rank(self.pg_in, j) if self.pg_in is not None else j,
                               group=self.pg_in)
                recv[off:off + rows[j]].copy_(full[:, self.p0:self.p0 + Pr])
                off += rows[j]
        sh = dict(x=recv, rows=rows, off=sum(rows[:self.rank]), n=N, work=work if async_op else None)
        return sh

    def prefetch(self, x_dep, slot=0):
        """Start the exchange for a batch a later step will use (its depth tensor must stay unchanged until then)."""
        self.ready[x_dep.data_ptr()] = self.shard_inputs(x_dep, slot, async_op=self.native_a2a)

    def inputs(self, x_dep):
        sh = self.ready.pop(x_dep.data_ptr(), None)
        if sh is None:
            sh = self.shard_inputs(x_dep)
        if sh["work"] is not None:
            sh["work"].wait()             # stream-level join, the host does not block
            sh["work"] = None
        assert sh["n"] == x_dep.shape[0]
        return sh

    # -- forward -----------------------------------------------------------------------------------------------
    def partial_forward(self, w, x_dep, ws):
        sh = self.inputs(x_dep)
        tot = sh["x"].shape[0]
        w.tp_in = sh
        w.tp_part = self._buf(("part", tot), (tot, self.eng.H))
        self.gemm(GEMM_NT, sh["x"], self.w, w.tp_part, ws=ws)

    def exchange_forward(self, w):
        dist.all_reduce(w.tp_part, group=self.pg)

    def summed(self, w):
        o = w.tp_in["off"]
        return w.tp_part[o:o + w.tp_in["n"]]

    # -- backward ----------------------------------------------------------------------------------------------
    def exchange_backward(self, w):
        sh = w.tp_in
        tot = sh["x"].shape[0]
        g = self._buf(("dpre", tot), (tot, self.eng.H))
        if self.native_gather and all(r == sh["n"] for r in sh["rows"]):
            try:                                             # one collective, no staging kernels on the critical path
                dist.all_gather_into_tensor(g, w.d_dep_pre, group=self.pg)
                w.tp_dpre_all = g
                return
            except (RuntimeError, NotImplementedError):
                self.native_gather = False
        g.zero_()
        g[sh["off"]:sh["off"] + sh["n"]].copy_(w.d_dep_pre)
        dist.all_reduce(g, group=self.pg)                    # a gather: every other rank contributed zeros here
        w.tp_dpre_all = g

    def wgrad(self, w, ws, adam=None):
        """adam (engine.backward(fused_adamw=...)): the owned columns are updated inside the GEMM's epilogue."""
        if adam is not None:
            self.gemm(GEMM_TN, w.tp_dpre_all, w.tp_in["x"], self.w, ws=ws, adam=dict(adam, m=self.m, v=self.v))
        else:
            self.gemm(GEMM_TN, w.tp_dpre_all, w.tp_in["x"], self.g, ws=ws)

    # -- replicated view (validation, checkpoints) -----------------------------------------------------------------
    def sync_full_weight(self):
        """All ranks end with the complete depth_projection.weight (each owned only its pixel columns)."""
        W = self.eng.arena.p("depth_projection.weight")
        tmp = torch.zeros_like(W)
        tmp[:, self.p0:self.p0 + self.Pr] = self.w
        dist.all_reduce(tmp, group=self.pg)
        W.copy_(tmp)


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class _BnScratch:
    """Stand-in for an nn.BatchNorm1d holder: the real weight, throw-away running buffers."""

    def __init__(self, bn):
        self.weight = bn.weight
        self.running_mean = torch.zeros_like(bn.running_mean)
        self.running_var = torch.ones_like(bn.running_var)
        self.num_batches_tracked = None


class SyncBatchNorm:
    """BatchNorm statistics of the BN-blend fuser (futr_safuser_batchnormalization.py:45-46) over the global batch.

    forward : every rank computes its local mean and M2 = sum (x - mean_local)^2 per channel with the same kernel as the
              one-GPU path (two-pass, so no E[x^2] - mean^2 cancellation), one all-gather of [2 modalities x 2 x C (+ row
              count)] floats, then Chan's parallel-variance combination in rank order (deterministic):
              mean = sum n_r mean_r / n,  M2 = sum M2_r + sum n_r (mean_r - mean)^2.  The running statistics are updated
              from the global moments (unbiased with the global count), as one process would.
    backward: dx = gamma rstd (dy - sum(dy)/n - xhat sum(dy xhat)/n) needs the two sums over the global batch: the local
              BatchNorm parameter gradients (which ARE those sums) are all-reduced as a 4 x C copy -- the arena keeps the
              local values for the ordinary gradient bucket -- and pre-scaled by n_local / n so that the kernel's own 1/n_local
              gives 1/n.
    All of it is enqueued (tiny device tensors, no host read), so it is captured with the step."""

    def __init__(self, dp):
        self.dp = dp
        self.scratch = None
        self.bufs = {}
        self.momentum = 0.1

    def forward(self, eng, w, mod):
        dp = self.dp
        if self.scratch is None:
            self.scratch = (_BnScratch(mod.bn_rgb), _BnScratch(mod.bn_depth))
        N, C = w.rgb.shape
        ops.bn_stats(w.rgb, w.dep, self.scratch[0], self.scratch[1], w.bn_mean, w.bn_rstd, w.bn_absg, True)
        key = (C, dp.world)
        if self.bufs.get("key") != key:
            dev = w.rgb.device
            self.bufs = dict(key=key, pack=torch.empty(4 * C + 1, dtype=torch.float32, device=dev),
                             allp=torch.empty(dp.world, 4 * C + 1, dtype=torch.float32, device=dev),
                             nfrac=torch.ones(1, dtype=torch.float32, device=dev))
        b = self.bufs
        ops.bn_sync_pack(w.bn_mean, w.bn_rstd, N, b["pack"])
        dp.all_gather_(b["allp"], b["pack"])
        ops.bn_sync_finalize(b["allp"], N, w.bn_mean, w.bn_rstd, mod.bn_rgb, mod.bn_depth, b["nfrac"], self.momentum)
        w.bn_nfrac = b["nfrac"]

    def backward_sums(self, eng, w):
        """-> [4, C]: global (d gamma_rgb, d beta_rgb, d gamma_depth, d beta_depth), scaled by n_local / n."""
        a = eng.arena
        t = torch.stack([a.g("fuser.bn_rgb.weight"), a.g("fuser.bn_rgb.bias"), a.g("fuser.bn_depth.weight"),
                         a.g("fuser.bn_depth.bias")])
        self.dp.all_reduce_(t)
        return t * w.bn_nfrac


class DataParallelStep:
    def __init__(self, engine, process_group=None, pixel_shard=False, equal_batches=True, input_group=None,
                 sync_bn=True):
        """sync_bn (BN-blend variant only): batch statistics over the GLOBAL batch, as the single-process reference sees it
        (False = per-rank statistics, what nn.DataParallel gives the reference)."""
        self.eng = engine
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # R3D_REHEARSE_DIST=1: run every exchange even with one rank (exercises the RCCL calls on a one-GPU box)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("R3D_REHEARSE_DIST") == "1")
        self.works = []
        a = engine.arena
        self.small = a.grads[a.bucket_small[0]:a.bucket_small[1]]
        self.big = a.grads[a.bucket_big[0]:a.bucket_big[1]]
        self._den = torch.zeros(1, dtype=torch.float32, device=a.grads.device)
        engine.grad_hook = self._on_stage
        engine.score_allreduce = self._scores
        self.comm_override = None         # RcclStep: exchanges issued from inside the step go through its communicator
        if getattr(engine, "bn", False) and sync_bn and self.active:
            engine.bn_sync = SyncBatchNorm(self)
        if self.active:
            engine.dur_den = self._den
        self.tp = None
        if pixel_shard and self.active:
            self.tp = PixelShardedDepth(engine, process_group, equal_batches, input_group)

    # -- exchanges issued from inside a step (scores, BatchNorm sums) --------------------------------------------
    def all_reduce_(self, t):
        if self.comm_override is not None:
            self.comm_override.all_reduce(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

    def all_gather_(self, out, inp):
        """out [world, ...inp.shape] <- every rank's inp."""
        if self.comm_override is not None:
            self.comm_override.all_gather(out, inp)
        elif dist.get_backend(self.pg) == "nccl":
            dist.all_gather_into_tensor(out, inp, group=self.pg)
        else:
            parts = [torch.empty_like(inp) for _ in range(self.world)]
            dist.all_gather(parts, inp, group=self.pg)
            out.copy_(torch.stack(parts))

    # -- 1. gradients ------------------------------------------------------------------------------------------
    def _on_stage(self, stage):
        if not self.active:
            return
        if stage == "big_ready" and self.tp is not None:
            return                        # that gradient is already complete on the rank that owns the columns
        buf = self.small if stage == "small_ready" else self.big
        if stage == "small_ready":
            buf = buf[:self._small_live()]
        if buf.numel():
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _small_live(self):
        """Floats of the small bucket that can be non-zero: pos_embedding [1, 2000, H] sits last in it and only its first
        S rows (the clip length of this step) receive a gradient -- the rest is exact zeros on every rank."""
        a, last = self.eng.arena, getattr(self.eng, "last", None)
        offs = getattr(a, "offsets", None)
        if not last or not offs or "pos_embedding" not in offs:
            return self.small.numel()
        o, n, shp = offs["pos_embedding"]
        if o + n != a.bucket_small[1]:
            return self.small.numel()
        return min(self.small.numel(), o - a.bucket_small[0] + last["w"].S * shp[-1])

    def wait_grads(self):
        for w in self.works:
            w.wait()
        self.works = []

    @property
    def grad_scale(self):
        return 1.0 / self.world

    # -- 2. duration denominator ---------------------------------------------------------------------------------
    def prepare_duration_denominator(self, target_dur, pad_idx, async_group=None):
        """Call before the step's loss kernel: den = (global count of non-pad duration targets) / world.
        async_group: run the all-reduce asynchronously on that communicator (under the input projections); the caller
        joins it with wait_duration_denominator() before the loss kernel is enqueued."""
        if not self.active:
            return
        if async_group is None or dist.get_backend(async_group) != "nccl":     # (stream-ordered async work is RCCL's)
            torch.mul((target_dur != pad_idx).sum().to(torch.float32).reshape(1), 1.0 / self.world, out=self._den)
            dist.all_reduce(self._den, op=dist.ReduceOp.SUM, group=self.pg)
            return
        # the count (4 tiny kernels) and its all-reduce run on a side stream: off the step's critical path
        if getattr(self, "_den_stream", None) is None:
            self._den_stream = torch.cuda.Stream(self._den.device)
        cur = torch.cuda.current_stream()
        self._den_stream.wait_stream(cur)
        with torch.cuda.stream(self._den_stream):
            torch.mul((target_dur != pad_idx).sum().to(torch.float32).reshape(1), 1.0 / self.world, out=self._den)
            self._den_work = dist.all_reduce(self._den, op=dist.ReduceOp.SUM, group=async_group, async_op=True)

    def wait_duration_denominator(self):
        w, self._den_work = getattr(self, "_den_work", None), None
        if w is not None:
            with torch.cuda.stream(self._den_stream):
                w.wait()
            torch.cuda.current_stream().wait_stream(self._den_stream)

    # -- 3. eval-mode selection scores ------------------------------------------------------------------------
    def _scores(self, sums, n_local):
        if not self.active:
            return float(n_local)
        cnt = torch.tensor([float(n_local)], dtype=torch.float64, device=sums.device)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.pg)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=self.pg)
        return float(cnt.item())

    def broadcast_parameters(self, src=0):
        """Same initial weights on every rank (nn.DataParallel replicates from device 0 each step)."""
        if self.active:
            dist.broadcast(self.eng.arena.params, src=src, group=self.pg)


class RcclStep:
    """The multi-GPU training step with its exchanges enqueued by RCCL on the step's own stream (r3d_amd/rccl.py), so that
    the whole step -- kernels and collectives -- is ONE linear hipGraph: no stream forks, no event joins inside it.

    What a step needs from the other ranks BEFORE it starts is staged one step ahead by stage(): the depth input's
    all-to-all (pixel-sharded mode) and the duration-loss denominator (1 float all-reduced).  stage() runs on its own
    stream / communicator (its own small graph), ordered against the step graphs only at graph boundaries, by events
    that are long complete when they are waited on -- so nothing of it sits on the critical path.

    Pixel-sharded mode, per step (W ranks, N frame rows per rank, H hidden, P pixels):
        stage: [N, P] -> all-to-all -> [W N, P/W] for the NEXT step (slot s^1);  its denominator
        run:   partial GEMM -> reduce-scatter [W N, H] (each rank keeps the sum of its own rows) -> ... -> losses ->
               backward -> all-gather d(depth_pre) [N, H] -> [W N, H] -> all-reduce of the replicated parameters' bucket
               -> weight-gradient GEMM on the owned columns -> AdamW (owned columns + replicated parameters)
    Replicated mode: both gradient buckets all-reduced in stream order (the small one before the depth weight-gradient
    GEMM, the large one after it).
    """

    def __init__(self, dp, comm, comm_side, lr, weight_decay, fuse_adam=False):
        self.dp, self.eng, self.tp = dp, dp.eng, dp.tp
        self.comm, self.side_comm = comm, comm_side
        self.lr, self.wd, self.fuse_adam = lr, weight_decay, fuse_adam and dp.tp is not None
        dev = self.eng.device
        self.den = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in range(2)]
        self.shards = {}

    def stage(self, x_dep2d, dur, pad_idx, slot):
        """(on the current stream) everything a step that will run from `slot` needs from the other ranks beforehand."""
        W = self.dp.world
        torch.mul((dur != pad_idx).sum().to(torch.float32).reshape(1), 1.0 / W, out=self.den[slot])
        self.side_comm.all_reduce(self.den[slot])
        if self.tp is not None:
            tp = self.tp
            # x_dep2d: [N, P] frame rows (re-laid out here, one 25.7 MB pass), or already [W, N, P/W] pixel-block-major --
            # the layout the input pipeline can produce for free at host-to-device time -- which is sent as it is
            blocked = x_dep2d.dim() == 3
            N = x_dep2d.shape[1] if blocked else x_dep2d.shape[0]
            recv = tp._buf(("recv", W * N, slot), (W * N, tp.Pr))
            send = x_dep2d if blocked else tp._buf(("send", N, slot), (W, N, tp.Pr))
            # One code path for every world size (a one-rank rehearsal captures a memcpy node for the contiguous copy_ and a
            # kernel node for the self-addressed all-to-all).  Round 1 replaced these two by multiply-by-one kernels at W == 1
            # after a segfault in capture_end, blaming "25.7 MB memcpy nodes"; tools/capture_probe.py (round 2) shows memcpy
            # nodes of 1-64 MB, on the capture stream or a side stream, alone or chained, and every one-rank RCCL collective
            # (captured as kernel / memcpy / nothing) instantiate and replay, and this very flow runs with the copies
            # restored -- the abort does not reproduce and was not caused by the node type.
            if not blocked:
                send.copy_(x_dep2d.view(N, W, tp.Pr).transpose(0, 1))
            self.side_comm.all_to_all(recv, send)
            self.shards[slot] = dict(x=recv, rows=[N] * W, off=tp.rank * N, n=N, work=None)

    def run(self, feats, depth, lab, dur, tgt, pad_idx, training, slot=0, lr=None, hyper=None, after_losses=None,
            stage_den=False, prefill_dropout=True):
        """Enqueue (or capture) one step whose staged inputs are in `slot`.
        stage_den: compute and all-reduce the loss denominator inside the step (a loop that only sees a batch when its
        step starts cannot stage it a step ahead); lr / hyper = (weight_decay, betas, eps): override the constructor's;
        after_losses(loss, counts): called where the loss kernel's outputs exist (epoch accumulators)."""
        eng, dp, tp = self.eng, self.dp, self.tp
        hook, eng.grad_hook = eng.grad_hook, None
        dp.comm_override = self.comm
        keep_defer, eng.defer_tail = eng.defer_tail, True     # forward -> losses -> backward back to back
        # nobody reads the loss statistics before the step's AdamW launch: their reduction rides there (engine.adamw)
        keep_red, eng.defer_loss_reduce = eng.defer_loss_reduce, after_losses is None
        wd, betas, eps = hyper if hyper is not None else (self.wd, (0.9, 0.999), 1e-8)
        lr = self.lr if lr is None else lr
        try:
            if stage_den:
                torch.mul((dur != pad_idx).sum().to(torch.float32).reshape(1), 1.0 / dp.world, out=self.den[slot])
                self.comm.all_reduce(self.den[slot])
            eng.dur_den = self.den[slot]
            if tp is not None:
                tp.ready[depth.reshape(depth.shape[0] * depth.shape[1], -1).data_ptr()] = self.shards[slot]
                eng.forward_begin(feats, depth, lab, "train", training)
                self.comm.reduce_scatter_inplace(eng._fw["w"].tp_part)
                eng.forward_finish()
            else:
                eng.forward(feats, depth, lab, "train", training)
            loss, counts = eng.losses(lab, tgt, dur, tick=True)
            if after_losses is not None:
                after_losses(loss, counts)
            adam = dict(lr=lr, weight_decay=wd, betas=betas, eps=eps, grad_scale=dp.grad_scale) if self.fuse_adam else None
            eng.prepare_fused_adamw(adam)
            eng.backward_main()
            w = eng.last["w"]
            small = dp.small[:dp._small_live()]
            with self.comm.group():                 # one submission: the gather the weight-gradient GEMM waits for and
                if tp is not None:                  # the replicated parameters' gradient bucket
                    g = tp._buf(("dpre", w.tp_part.shape[0]), tuple(w.tp_part.shape))
                    self.comm.all_gather(g, w.d_dep_pre)
                    w.tp_dpre_all = g
                if small.numel():
                    self.comm.all_reduce(small)
            eng.backward_depth_wgrad()
            if tp is None:
                self.comm.all_reduce(dp.big)
            eng.adamw(lr, wd, betas=betas, eps=eps, grad_scale=dp.grad_scale, ticked=True, skip_depth=self.fuse_adam,
                      prefill_dropout=prefill_dropout)
        finally:
            eng.grad_hook = hook
            dp.comm_override = None
            eng.defer_tail, eng.defer_loss_reduce = keep_defer, keep_red
