"""Drop-in for the reference's ``train/train_proposed_depth.py``: ``train(args, model, train_loader, optimizer,
scheduler, criterion, model_save_path, pad_idx, device, val_loader, seed)`` and ``validate(model, val_loader,
criterion, pad_idx, device)`` with the reference's signatures, skip rules, loss composition, prints and checkpoint
names (train_proposed_depth.py:52-108, 110-253) -- but each batch is ONE enqueue of the fused HIP step (forward + 3
losses + backward + AdamW) with no device->host sync; epoch statistics are read back once per epoch.

Kept quirks of the reference (each can be checked against the cited line):
  * batches with fewer than 8 clips are skipped (:148)  [--min_batch];
  * model.eval() set by validate() (:53) is never undone, so dropout is active in epoch 0 only  [--restore_train_mode];
  * validate() compares the normalised duration with the UNMASKED target (:98-99) and skips the seg loss;
  * checkpoints: seed_{seed}_checkpoint{epoch}.ckpt and seed_{seed}_best.ckpt on accuracy improvement (:237-249).
"""
import os

import torch
import torch.distributed as dist

from .model.futr_safuser_tokenfusion import FUTR
from .model.futr_unsupervised_depth import FUTR as FUTRDepthQuery
from .optim import FlatAdamW
from .parallel import DataParallelStep


def _unwrap(model):
    m = model
    while hasattr(m, "module") and not isinstance(m, (FUTR, FUTRDepthQuery)):
        m = m.module
    if not isinstance(m, (FUTR, FUTRDepthQuery)):
        raise TypeError("r3d_amd.train_proposed_depth drives r3d_amd.model.futr_safuser_tokenfusion.FUTR (or its "
                        "BN-blend subclass) and r3d_amd.model.futr_unsupervised_depth.FUTR")
    return m


def get_last_non_padding_labels(past_label, pad_value):
    """train_proposed_depth.py:28-50, without the per-clip python loop / host sync."""
    S = past_label.size(1)
    pos = torch.arange(S, device=past_label.device).expand_as(past_label)
    last = torch.where(past_label != pad_value, pos, torch.full_like(pos, -1)).max(dim=1).values
    out = past_label.gather(1, last.clamp_min(0).unsqueeze(1)).squeeze(1)
    return torch.where(last >= 0, out, torch.full_like(out, pad_value))


def weighted_accuracy(pred, gold, pad_idx, t_n_labels, weight_same=1.0, weight_different=10.0):
    """train_proposed_depth.py:9-26 (the weight cancels in the ratio; kept for signature parity)."""
    pred = pred.max(1)[1]
    mask = gold.ne(pad_idx)
    total = int(mask.sum())
    return float((pred.eq(gold) & mask).sum()) / total if total > 0 else 0


def _to_dev(data, device):
    features, depth_features, past_label, trans_dur_future, trans_future_target = data
    return (features.to(device=device, dtype=torch.float32), depth_features.to(device=device, dtype=torch.float32),
            past_label.to(device).long().contiguous(), trans_dur_future.to(device=device, dtype=torch.float32).contiguous(),
            trans_future_target.to(device).long().contiguous())


class _GraphedSteps:
    """Replays the fused training step (forward + losses + backward + AdamW + epoch accumulators) as ONE hipGraph per
    batch shape.  Enqueued launch by launch from Python the 36-launch step is bound by the host (measured 1.68 ms/step
    against 0.29 ms replayed), so the loop keeps static device buffers per (B, S) shape, copies each batch into them
    (device-to-device, or straight from the loader when it is given these buffers) and replays.  The first step of a
    shape runs eagerly (allocations, planner), the second one is captured; lr lives in device memory and may change
    between replays, the other AdamW hyper-parameters are part of the capture (a change re-captures)."""

    def __init__(self, eng, acc_loss, acc_cnt, dp=None, pad_idx=None, rs=None):
        """dp (replicated data parallel): with rs (parallel.RcclStep: RCCL enqueued on the launch stream) the step stays
        ONE graph, exchanges included; otherwise it becomes three graphs around the two torch.distributed all-reduces:
        [forward, losses, backward] -> small bucket (async, under the next graph) -> [depth weight gradient] -> big bucket
        -> [AdamW]."""
        self.eng, self.acc_loss, self.acc_cnt = eng, acc_loss, acc_cnt
        self.dp, self.pad_idx, self.rs = dp, pad_idx, rs
        self.shapes = {}

    def step(self, batch, lr, hyper, training):
        eng = self.eng
        key = tuple(tuple(t.shape) for t in batch) + (bool(training), float(eng.erank_weight))
        st = self.shapes.get(key)
        if st is None:
            st = self.shapes[key] = dict(buf=[torch.empty_like(t) for t in batch], seen=0, graph=None, hyper=None)
        for dst, src in zip(st["buf"], batch):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        eng.set_lr(lr)
        if self.dp is not None and self.rs is None:
            return self._step_dp(st, lr, hyper, training)
        if st["graph"] is not None and st["hyper"] == hyper:
            st["graph"].replay()
            return
        eng._drop_ready = None
        if self.rs is not None:
            run = lambda: self.rs.run(*st["buf"], self.pad_idx, training, slot=0, lr=lr, hyper=hyper,     # noqa: E731
                                      after_losses=self._accumulate, stage_den=True, prefill_dropout=False)
        else:
            run = lambda: self._enqueue(st["buf"], lr, hyper, training)     # noqa: E731
        if st["seen"] == 0 or st["hyper"] not in (None, hyper):
            run()                                                           # eager: sizes every workspace
            st["seen"], st["hyper"], st["graph"] = 1, hyper, None
            return
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            run()
        st["graph"], st["hyper"] = g, hyper
        g.replay()

    def _accumulate(self, loss, counts):
        self.acc_loss += loss
        self.acc_cnt += counts

    def _step_dp(self, st, lr, hyper, training):
        eng, dp = self.eng, self.dp
        feats, depth, lab, dur, tgt = st["buf"]
        wd, betas, eps = hyper
        dp.prepare_duration_denominator(dur, self.pad_idx)

        def part1():
            eng.forward(feats, depth, lab, "train", training=training)
            loss, counts = eng.losses(lab, tgt, dur, tick=True)
            eng.backward_main()
            self.acc_loss += loss
            self.acc_cnt += counts

        def part3():
            eng.adamw(lr, wd, betas=betas, eps=eps, grad_scale=dp.grad_scale, ticked=True)
        eng._drop_ready = None
        hook, eng.grad_hook = eng.grad_hook, None            # the exchanges are issued here, between the graphs
        try:
            if st["graph"] is None or st["hyper"] != hyper:
                if st["seen"] == 0 or st["hyper"] not in (None, hyper):
                    g1 = g2 = g3 = None                          # first step of this shape: eager
                    st["seen"], st["hyper"], st["graph"] = 1, hyper, None
                else:
                    torch.cuda.synchronize()
                    g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1):
                        part1()
                    with torch.cuda.graph(g2):
                        eng.backward_depth_wgrad()
                    with torch.cuda.graph(g3):
                        part3()
                    st["graph"], st["hyper"] = (g1, g2, g3), hyper
            else:
                g1, g2, g3 = st["graph"]
            (g1.replay if g1 is not None else part1)()
            dp._on_stage("small_ready")
            (g2.replay if g2 is not None else eng.backward_depth_wgrad)()
            dp._on_stage("big_ready")
            dp.wait_grads()
            (g3.replay if g3 is not None else part3)()
        finally:
            eng.grad_hook = hook

    def _enqueue(self, buf, lr, hyper, training):
        eng = self.eng
        feats, depth, lab, dur, tgt = buf
        wd, betas, eps = hyper
        # forward -> losses -> backward back to back: the decoder tail, the losses and the tail's backward are one launch
        # (defer_tail), and the loss statistics -- with their epoch sums -- are reduced by one workgroup of the AdamW launch
        fused = hasattr(eng, "defer_loss_reduce")            # (the depth-as-query engine keeps its separate launches)
        if fused:
            keep = (eng.defer_tail, eng.defer_loss_reduce, eng.loss_acc)
            eng.defer_tail, eng.defer_loss_reduce, eng.loss_acc = True, True, (self.acc_loss, self.acc_cnt)
        try:
            eng.forward(feats, depth, lab, "train", training=training)
            loss, counts = eng.losses(lab, tgt, dur, tick=True)
            folded = getattr(eng, "_loss_pending", None) is not None
        finally:
            if fused:
                eng.defer_tail, eng.defer_loss_reduce, eng.loss_acc = keep
        fuse = bool(fused and eng.depth_adamw_fusable())         # (wide / long shapes: AdamW of the depth weight in its wgrad kernel)
        eng.backward(fused_adamw=dict(lr=lr, weight_decay=wd, betas=betas, eps=eps) if fuse else None, adamw_next=True)
        eng.adamw(lr, wd, betas=betas, eps=eps, ticked=True, skip_depth=fuse)     # (no dropout prefill: every captured step generates its
        if not folded:                                            #  own masks, so graphs of different shapes can interleave)
            self.acc_loss += loss
            self.acc_cnt += counts


class _ShardedSteps:
    """--pixel_shard training steps on parallel.RcclStep with a ONE-BATCH LOOK-AHEAD: while step t runs (one hipGraph,
    RCCL exchanges captured in it), batch t + 1 is copied into the other slot's static buffers on a side stream and its
    depth all-to-all + loss denominator are staged there (a second small graph), so nothing step t + 1 needs from the other
    ranks sits on its critical path.  The reference has no counterpart (nn.DataParallel scatters inside the step,
    main_darai.py:133; the loop sees one batch at a time, train_proposed_depth.py:127-138).

    Requirements of the static exchange: every rank feeds batches of the same shape in the same order (a
    DistributedSampler with drop_last); a batch that was not pre-staged (first of an epoch, first of a new shape) is
    staged on the spot.  Ordering between the streams is as in bench.py: the step waits for its slot's staging event (long
    complete); the staging of slot s^1 may only overwrite it when the step that last read it is done -- that
    dependency is kept on the host (an event synchronize of step t - 1 after step t has been queued)."""

    def __init__(self, eng, dp, rs, acc_loss, acc_cnt, pad_idx, use_graphs=True):
        self.eng, self.dp, self.rs, self.pad_idx = eng, dp, rs, pad_idx
        self.acc_loss, self.acc_cnt = acc_loss, acc_cnt
        self.use_graphs = use_graphs
        self.shapes = {}
        self.side = torch.cuda.Stream(eng.device)
        self.t = 0
        self.ev_step = [torch.cuda.Event() for _ in range(4)]
        for e in self.ev_step:
            e.record()

    def _accumulate(self, loss, counts):
        self.acc_loss += loss
        self.acc_cnt += counts

    @staticmethod
    def _key(batch, training):
        return tuple((tuple(t.shape), str(t.dtype)) for t in batch) + (bool(training),)

    def _state(self, batch, training):
        key = self._key(batch, training)
        st = self.shapes.get(key)
        if st is None:
            dev = self.eng.device
            mk = lambda: [torch.empty(t.shape, dtype=t.dtype, device=dev) for t in batch]      # noqa: E731
            st = self.shapes[key] = dict(buf=[mk(), mk()], slot=0, staged=[None, None], runs=[0, 0], G=[None, None],
                                         S=[None, None], ev_side=[torch.cuda.Event(), torch.cuda.Event()], hyper=None,
                                         feats=torch.empty(batch[0].shape, dtype=batch[0].dtype, device=dev))
            for e in st["ev_side"]:
                e.record()
        return st

    def _stage(self, st, s):
        buf = st["buf"][s]
        x2d = buf[1].reshape(buf[1].shape[0] * buf[1].shape[1], -1)
        self.rs.stage(x2d, buf[3], self.pad_idx, s)

    def _fill(self, st, s, batch, token):
        """(current stream = the side stream) batch -> slot s buffers (host-to-device or device-to-device), then its staging."""
        assert torch.cuda.current_stream() == self.side
        for dst, src in zip(st["buf"][s], batch):
            dst.copy_(src, non_blocking=True)
        if self.use_graphs and st["S"][s] is None and st["staged"][s] is not None:
            # second staging of this slot: capture it (the first one sized the exchange buffers)
            g = torch.cuda.CUDAGraph()
            torch.cuda.current_stream().synchronize()
            with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
                self._stage(st, s)
            st["S"][s] = g
        if st["S"][s] is not None:
            st["S"][s].replay()
        else:
            self._stage(st, s)
        st["staged"][s] = token

    def step(self, cur, cur_token, nxt, nxt_token, lr, hyper, training):
        """cur / nxt: 5-tuples of tensors (host or device; nxt may be None); tokens identify batches (any hashable)."""
        eng, rs = self.eng, self.rs
        st = self._state(cur, training)
        s = st["slot"]
        main = torch.cuda.current_stream()
        ev_pre = torch.cuda.Event()
        ev_pre.record(main)                                 # everything the caller produced so far (device-resident batches)
        eng.set_lr(lr)
        if st["staged"][s] != cur_token:                    # not pre-staged: first batch of an epoch / of this shape
            # (ragged loaders -- pad_sequence pads to the batch's longest clip -- come here whenever the shape changes.)
            # Staged on the SIDE stream like every look-ahead: the staging may be captured on this visit, and a hipGraph
            # cannot be captured on the default stream, where train() runs
            self.ev_step[(self.t - 1) % 4].synchronize()    # (the step that last read this slot is long done; be safe)
            self.side.wait_event(ev_pre)
            with torch.cuda.stream(self.side):
                self._fill(st, s, cur, cur_token)
                st["ev_side"][s].record(self.side)
        main.wait_event(st["ev_side"][s])
        buf = st["buf"][s]

        def run():
            # the RGB features go through ONE static buffer: the engine's grouped weight-gradient launch keeps their
            # address in a device-side descriptor table (ops.GemmGroup), which a captured step cannot re-point per slot
            st["feats"].copy_(buf[0])
            rs.run(st["feats"], *buf[1:], self.pad_idx, training, slot=s, lr=lr, hyper=hyper, after_losses=self._accumulate,
                   stage_den=False, prefill_dropout=False)
        eng._drop_ready = None
        if not self.use_graphs:
            run()
        elif st["G"][s] is not None and st["hyper"] == hyper:
            st["G"][s].replay()
        elif st["runs"][s] == 0 or st["hyper"] not in (None, hyper):
            run()                                           # eager: sizes every workspace of this slot
            st["G"] = [None, None]
            st["hyper"] = hyper
        else:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                run()
            st["G"][s] = g
            g.replay()
        st["runs"][s] += 1
        self.ev_step[self.t % 4].record(main)
        # ---- look-ahead: the next batch into the other slot, on the side stream, under this step
        if nxt is not None and self._key(nxt, training) == self._key(cur, training):
            self.ev_step[(self.t - 1) % 4].synchronize()    # slot s^1 was last read by step t - 1
            self.side.wait_event(ev_pre)                    # (not the step just queued: the staging runs under it)
            with torch.cuda.stream(self.side):
                self._fill(st, s ^ 1, nxt, nxt_token)
                st["ev_side"][s ^ 1].record(self.side)
        st["slot"] = s ^ 1
        self.t += 1

    def drain(self):
        """End of an epoch: nothing staged is left pending on the side stream."""
        self.side.synchronize()


def validate(model, val_loader, criterion, pad_idx, device):
    core = _unwrap(model)
    model.eval()
    eng = core.engine()
    val_loss = 0.0
    val_class_correct = 0
    val_class_total = 0
    val_seg_correct = 0
    val_seg_total = 0
    val_weighted_accuracy_total = 0
    with torch.no_grad():
        for data in val_loader:
            if data is None:
                continue
            features, depth_features, past_label, trans_dur_future, trans_future_target = _to_dev(data, eng.device)
            out = eng.forward(features, depth_features, past_label, "val", training=False, need_grad=False)
            loss, counts = eng.losses(past_label, trans_future_target, trans_dur_future, with_grad=False, val_mode=True)
            lv, cv = loss.cpu(), counts.cpu()                       # one readback per validation clip
            val_loss += float(lv[1] + lv[2])                        # action CE + duration (:86,100)
            val_class_correct += int(cv[2])
            val_class_total += int(cv[3])
            val_weighted_accuracy_total += weighted_accuracy(
                out["action"].reshape(-1, out["action"].size(-1)), trans_future_target.view(-1), pad_idx,
                get_last_non_padding_labels(past_label, pad_idx))
    val_loss /= len(val_loader)
    val_accuracy = val_class_correct / val_class_total if val_class_total else 0
    val_seg_accuracy = val_seg_correct / val_seg_total if val_seg_total else 0
    val_weighted_accuracy = val_weighted_accuracy_total / len(val_loader)
    print(f"Validation Loss: {val_loss:.3f}, Class Accuracy: {val_accuracy:.3f}, Segmentation Accuracy: "
          f"{val_seg_accuracy:.3f}, Weighted Accuracy: {val_weighted_accuracy:.3f}")
    return val_loss, val_accuracy, val_weighted_accuracy


def train(args, model, train_loader, optimizer, scheduler, criterion, model_save_path, pad_idx, device, val_loader, seed):
    core = _unwrap(model)
    model.to(device)
    model.train()
    eng = core.engine()
    eng.defer_tail = True       # every step here is forward -> losses -> backward: tail forward, losses and tail backward
                                # run as one launch (validate()'s forwards carry no gradient workspace and are unaffected)
    dp = None
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1
                                                          or os.environ.get("R3D_REHEARSE_DIST") == "1"):
        # --pixel_shard: depth_projection tensor-parallel over pixels (parallel.PixelShardedDepth); per-rank batch sizes may
        # differ at the end of an epoch, so the row counts are exchanged every step
        dp = DataParallelStep(eng, pixel_shard=getattr(args, "pixel_shard", False), equal_batches=False)
    if dp is not None:
        dp.broadcast_parameters()
    is_main = dp is None or dp.rank == 0
    min_batch = getattr(args, "min_batch", 8)
    erank_every = getattr(args, "erank_every", 0)
    eng.erank_weight = float(getattr(args, "erank_weight", 0.0))
    print("Training Start")
    best_val_loss = float("inf")
    best_val_acc = 0
    best_weight_acc = 0
    acc_loss = torch.zeros(4, dtype=torch.float64, device=eng.device)
    acc_cnt = torch.zeros(4, dtype=torch.int64, device=eng.device)
    # one-GPU steps with the fused optimiser replay as hipGraphs (--no_graph_steps / args.graph_steps=False: eager)
    graphed = sharded = None
    rs = None
    if dp is not None and dp.active and getattr(args, "graph_steps", True) and not getattr(args, "torch_collectives", False):
        g0 = optimizer.param_groups[0]
        make_comm = getattr(args, "comm_factory", None)     # (tests: a torch.distributed stand-in for the RCCL binding)
        if make_comm is not None or dist.get_backend() == "nccl":
            try:                                    # RCCL on the launch stream: the data-parallel step stays one graph
                from .parallel import RcclStep
                if make_comm is None:
                    from .rccl import RcclComm
                    make_comm = RcclComm
                rs = RcclStep(dp, make_comm(), make_comm(), g0["lr"], g0["weight_decay"])
                if dp.rank == 0:
                    print("Data-parallel step: RCCL on the launch stream, one hipGraph per step")
            except Exception as e:                  # noqa: BLE001  (no librccl beside torch, communicator refused, ...)
                print(f"RCCL step unavailable ({type(e).__name__}: {e}); using torch.distributed all-reduces")
                rs = None
    if getattr(args, "graph_steps", True) and (dp is None or dp.tp is None):
        graphed = _GraphedSteps(eng, acc_loss, acc_cnt, dp, pad_idx, rs)
    elif rs is not None and dp.tp is not None and isinstance(optimizer, FlatAdamW):
        # --pixel_shard: the sharded one-graph step with a one-batch look-ahead (equal batch shapes on every rank)
        sharded = _ShardedSteps(eng, dp, rs, acc_loss, acc_cnt, pad_idx, use_graphs=getattr(args, "sharded_graphs", True))
        dp.tp.equal_batches = True
    for epoch in range(args.epochs):
        acc_loss.zero_()
        acc_cnt.zero_()
        n_steps, i = 0, -1
        if sharded is not None:
            # one-batch look-ahead over the VALID batches (the reference's skip rules, :128,148, applied first)
            def valid(it):
                for j, data in it:
                    if data is None or len(data[0]) < min_batch:
                        continue
                    yield j, data
            it = valid(enumerate(train_loader))
            cur = next(it, None)
            last_i = -1
            while cur is not None:
                nxt = next(it, None)
                g = optimizer.param_groups[0]
                prep = lambda d: (d[0].float().contiguous(), d[1].float().contiguous(), d[2].long().contiguous(),     # noqa: E731
                                  d[3].float().contiguous(), d[4].long().contiguous())
                sharded.step(prep(cur[1]), (epoch, cur[0]), prep(nxt[1]) if nxt is not None else None,
                             (epoch, nxt[0]) if nxt is not None else None, g["lr"],
                             (g["weight_decay"], tuple(g["betas"]), g["eps"]), model.training)
                n_steps += 1
                last_i = cur[0]
                cur = nxt
            sharded.drain()
            i = max(last_i, len(train_loader) - 1) if hasattr(train_loader, "__len__") else last_i
        for i, data in (enumerate(train_loader) if sharded is None else ()):
            if data is None:
                continue
            features, depth_features, past_label, trans_dur_future, trans_future_target = _to_dev(data, eng.device)
            if len(features) < min_batch:
                continue
            g = optimizer.param_groups[0]
            if graphed is not None and isinstance(optimizer, FlatAdamW):
                graphed.step([features.contiguous(), depth_features.contiguous(), past_label.contiguous(),
                              trans_dur_future.contiguous(), trans_future_target.contiguous()], g["lr"],
                             (g["weight_decay"], tuple(g["betas"]), g["eps"]), model.training)
                n_steps += 1
                if erank_every and n_steps % erank_every == 0:
                    from .erank import effective_rank
                    print("effective rank of fused tokens: %.3f" % float(effective_rank(eng.last["w"].fused)))
                continue
            if dp is not None:
                dp.prepare_duration_denominator(trans_dur_future, pad_idx)
            eng.forward(features, depth_features, past_label, "train", training=model.training)
            fused_opt = isinstance(optimizer, FlatAdamW)
            loss, counts = eng.losses(past_label, trans_future_target, trans_dur_future, tick=fused_opt)
            fuse = bool(fused_opt and dp is None and hasattr(eng, "depth_adamw_fusable") and eng.depth_adamw_fusable())
            eng.backward(fused_adamw=dict(lr=g["lr"], weight_decay=g["weight_decay"], betas=g["betas"], eps=g["eps"])
                         if fuse else None, adamw_next=fused_opt and dp is None)
            if dp is not None:
                dp.wait_grads()
            if fused_opt:
                eng.adamw(g["lr"], g["weight_decay"], betas=g["betas"], eps=g["eps"],
                          grad_scale=dp.grad_scale if dp is not None else 1.0, ticked=True, prefill_dropout=True,
                          skip_depth=fuse)
            else:                                   # any other torch optimiser: expose the arena gradients to it
                if dp is not None:
                    eng.arena.grads.mul_(dp.grad_scale)
                eng.arena.attach_grads(core.named_parameters())
                optimizer.step()
                if eng.last["drop"]:
                    eng.drop_offset.add_(1)
            acc_loss += loss
            acc_cnt += counts
            n_steps += 1
            if erank_every and n_steps % erank_every == 0:
                from .erank import effective_rank
                print("effective rank of fused tokens: %.3f" % float(effective_rank(eng.last["w"].fused)))
        lsum, csum = acc_loss.cpu(), acc_cnt.cpu()                  # the single device->host read of the epoch
        denom = i + 1                                                # the reference divides by (i+1), skipped or not (:218)
        epoch_loss = float(lsum[3]) / denom if denom else 0.0
        print("Epoch [", (epoch + 1), "/", args.epochs, "] Loss : %.3f" % epoch_loss)
        if args.anticipate:
            accuracy = int(csum[2]) / int(csum[3]) if int(csum[3]) else 0.0
            print("Training Acc :%.3f" % accuracy, "CE loss :%.3f" % (float(lsum[1]) / denom if denom else 0.0))
            if args.task == "long":
                print("dur loss: %.5f" % (float(lsum[2]) / denom if denom else 0.0))
        scheduler.step()
        if dp is not None and dp.tp is not None:
            dp.tp.sync_full_weight()                                 # validation and checkpoints see the complete weight
        val_loss, val_acc, weight_acc = validate(model, val_loader, criterion, pad_idx, device)
        if getattr(args, "restore_train_mode", False):
            model.train()
        if (val_acc > best_val_acc or weight_acc > best_weight_acc) and is_main:
            best_val_loss, best_val_acc, best_weight_acc = val_loss, val_acc, weight_acc
            save_path = os.path.join(model_save_path)
            save_file = os.path.join(save_path, "seed_" + str(seed) + "_checkpoint" + str(epoch) + ".ckpt")
            torch.save(model.state_dict(), save_file)
            best_save_file = os.path.join(save_path, "seed_" + str(seed) + "_best.ckpt")
            if os.path.exists(best_save_file):
                os.remove(best_save_file)
            torch.save(model.state_dict(), best_save_file)
            print(f"Best model saved with validation loss: {best_val_loss:.3f}")
    eng.defer_tail = False
    return model
