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
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items clips for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DataParallelStep:
    def __init__(self, engine, process_group=None):
        self.eng = engine
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.works = []
        a = engine.arena
        self.small = a.grads[a.bucket_small[0]:a.bucket_small[1]]
        self.big = a.grads[a.bucket_big[0]:a.bucket_big[1]]
        self._den = torch.zeros(1, dtype=torch.float32, device=a.grads.device)
        engine.grad_hook = self._on_stage
        engine.score_allreduce = self._scores
        if self.world > 1:
            engine.dur_den = self._den

    # -- 1. gradients ------------------------------------------------------------------------------------------
    def _on_stage(self, stage):
        if self.world == 1:
            return
        buf = self.small if stage == "small_ready" else self.big
        if buf.numel():
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def wait_grads(self):
        for w in self.works:
            w.wait()
        self.works = []

    @property
    def grad_scale(self):
        return 1.0 / self.world

    # -- 2. duration denominator ---------------------------------------------------------------------------------
    def prepare_duration_denominator(self, target_dur, pad_idx):
        """Call before the step's loss kernel: den = (global count of non-pad duration targets) / world."""
        if self.world == 1:
            return
        self._den.copy_((target_dur != pad_idx).sum().to(torch.float32).reshape(1))
        dist.all_reduce(self._den, op=dist.ReduceOp.SUM, group=self.pg)
        self._den.div_(self.world)

    # -- 3. eval-mode selection scores ------------------------------------------------------------------------
    def _scores(self, sums, n_local):
        if self.world == 1:
            return float(n_local)
        cnt = torch.tensor([float(n_local)], dtype=torch.float64, device=sums.device)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.pg)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=self.pg)
        return float(cnt.item())

    def broadcast_parameters(self, src=0):
        """Same initial weights on every rank (nn.DataParallel replicates from device 0 each step)."""
        if self.world > 1:
            dist.broadcast(self.eng.arena.params, src=src, group=self.pg)
