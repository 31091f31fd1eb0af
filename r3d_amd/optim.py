"""Optimiser and schedule of the reference's entry scripts (main_darai.py:135-137) for the flat parameter arena.

FlatAdamW          torch.optim.AdamW(model.parameters(), lr, weight_decay=wd) semantics as ONE
                   HIP launch over the engine's [params | grads | exp_avg | exp_avg_sq] arenas.  It is a
                   torch.optim.Optimizer, so schedulers that rewrite param_groups[i]['lr'] keep working.
LinearWarmupCosineAnnealingLR
                   restatement of pl_bolts 0.3.4's scheduler (pl_bolts is not installed here and is not part of
                   /root/reference -> PARITY UNPINNED); defaults warmup_start_lr=0, eta_min=0.
"""
import math

import torch


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdamW drives one flat arena: pass a single parameter group (as main_darai.py:135 does)")

    def _engine(self):
        engines = {id(e): e for e in (getattr(p, "_r3d_engine", lambda: None)() for p in self.param_groups[0]["params"])
                   if e is not None}
        if len(engines) != 1:
            raise RuntimeError("FlatAdamW needs the parameters of exactly one r3d_amd FUTR whose engine exists "
                               "(run a forward on the GPU first)")
        return next(iter(engines.values()))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self._engine()
        g = self.param_groups[0]
        # gradients produced through autograd (the drop-in route) live outside the arena: bring them in
        for n, p in eng.module.named_parameters():
            if p.grad is not None and n in eng.arena.offsets and eng.arena.offsets[n][0] < eng.arena.n_live:
                tgt = eng.arena.g(n)
                if p.grad.data_ptr() != tgt.data_ptr():
                    tgt.copy_(p.grad)
        eng.adamw(g["lr"], g["weight_decay"], betas=g["betas"], eps=g["eps"], grad_scale=grad_scale)
        return loss


class LinearWarmupCosineAnnealingLR(torch.optim.lr_scheduler.LRScheduler):
    """lr(e) = warmup_start_lr + e*(base-warmup_start_lr)/(warmup_epochs-1) for e < warmup_epochs, then
    eta_min + (base-eta_min)*(1+cos(pi*(e-warmup)/(max-warmup)))/2.  Stepped once per epoch
    (train/train_proposed_depth.py:233); epoch 0 therefore runs at warmup_start_lr = 0."""

    def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=0.0, last_epoch=-1):
        self.warmup_epochs, self.max_epochs = warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        super().__init__(optimizer, last_epoch)

    def _lr_at(self, e, base):
        if self.warmup_epochs > 0 and e < self.warmup_epochs:
            if self.warmup_epochs == 1:
                return base
            return self.warmup_start_lr + e * (base - self.warmup_start_lr) / (self.warmup_epochs - 1)
        t, T = e - self.warmup_epochs, max(1, self.max_epochs - self.warmup_epochs)
        return self.eta_min + 0.5 * (base - self.eta_min) * (1.0 + math.cos(math.pi * t / T))

    def get_lr(self):
        return [self._lr_at(self.last_epoch, b) for b in self.base_lrs]
