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
        eng.forward(feats, depth, lab, "train", training=training)
        loss, counts = eng.losses(lab, tgt, dur, tick=True)
        eng.backward()
        eng.adamw(lr, wd, betas=betas, eps=eps, ticked=True)     # (no dropout prefill: every captured step generates its
        self.acc_loss += loss                                     #  own masks, so graphs of different shapes can interleave)
        self.acc_cnt += counts


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
    graphed = None
    if getattr(args, "graph_steps", True) and (dp is None or dp.tp is None):
        rs = None
        if dp is not None and dp.active and dist.get_backend() == "nccl" and not getattr(args, "torch_collectives", False):
            try:                                    # RCCL on the launch stream: the data-parallel step stays one graph
                from .parallel import RcclStep
                from .rccl import RcclComm
                g0 = optimizer.param_groups[0]
                rs = RcclStep(dp, RcclComm(), RcclComm(), g0["lr"], g0["weight_decay"])
                if dp.rank == 0:
                    print("Data-parallel step: RCCL on the launch stream, one hipGraph per step")
            except Exception as e:                  # noqa: BLE001  (no librccl beside torch, communicator refused, ...)
                print(f"RCCL step unavailable ({type(e).__name__}: {e}); using torch.distributed all-reduces")
                rs = None
        graphed = _GraphedSteps(eng, acc_loss, acc_cnt, dp, pad_idx, rs)
    for epoch in range(args.epochs):
        acc_loss.zero_()
        acc_cnt.zero_()
        n_steps, i = 0, -1
        for i, data in enumerate(train_loader):
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
            eng.backward()
            if dp is not None:
                dp.wait_grads()
            if fused_opt:
                eng.adamw(g["lr"], g["weight_decay"], betas=g["betas"], eps=g["eps"],
                          grad_scale=dp.grad_scale if dp is not None else 1.0, ticked=True, prefill_dropout=True)
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
