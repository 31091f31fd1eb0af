"""Host-side sequencing of the token-fusion training step on one MI355X.

The engine owns (a) flat fp32 arenas [params | grads | exp_avg | exp_avg_sq] whose slices back the nn.Parameters of
r3d_amd.model.futr_safuser_tokenfusion.FUTR, and (b) per-shape activation workspaces; it enqueues the HIP kernels of
libr3d_hip.so (r3d_amd.ops) in forward / backward order on torch's current stream.  No ATen math runs in the step:
torch only allocates buffers, provides the stream (and hipGraph capture) and the RCCL process group.

Forward restates FUTR.forward (model/futr_safuser_tokenfusion.py:164-239) with
  * rows ordered (clip, frame) b-major instead of the reference's seq-first [S,B,H];
  * the fuser's masked 2-token attention in closed form: softmax([[-inf,s],[s,-inf]]) == [[0,1],[1,0]] exactly, so
    attention == swap of V between the modality tokens and the Q/K projections receive exactly zero gradient
    (checked against the full attention of the oracle);
  * fc and fc_len evaluated as one [K+1, H] GEMM (their weights are adjacent in the arena).
Backward is the hand-derived adjoint of the same sequence (the reference gets it from autograd,
train/train_proposed_depth.py:214).
"""
import math

import torch

from . import ops
from ._lib import GEMM_NT, GEMM_NN, GEMM_TN

LIVE_PREFIXES = ("input_embed.", "depth_projection.", "depth_layernorm.", "pos_embedding", "query_embed.",
                 "fuser.blocks.", "fuser.norm.", "transformer.decoder.", "fc_seg.", "fc.", "fc_len.")
EXCLUDE_CLASS_IDX = 47          # hard-coded at train/train_proposed_depth.py:181,195
DROP_P = 0.1                    # every nn.Dropout on the path (futr_safuser_tokenfusion.py:26; transformer.py:22)


BN_LIVE_PREFIXES = ("fuser.alpha", "fuser.bn_rgb.", "fuser.bn_depth.")    # the BN-blend variant's extra trainable parameters


def is_live(name):
    """Parameters that receive a gradient in the reference step; the rest keep grad=None."""
    return name.startswith(LIVE_PREFIXES)


class ParamArena:
    """Flat arenas; live parameters first (AdamW touches only that prefix), depth_projection.weight last among them
    so that everything else forms one contiguous all-reduce bucket that is ready before the big weight gradient."""

    def __init__(self, named_params, device, extra_live=()):
        named = list(named_params)
        self.is_live = lambda n: is_live(n) or (bool(extra_live) and n.startswith(tuple(extra_live)))
        live = [(n, p) for n, p in named if self.is_live(n)]
        dead = [(n, p) for n, p in named if not self.is_live(n)]

        def key(item):
            n, p = item
            if n == "depth_projection.weight":
                return (3, 0)
            if n in ("fc.weight", "fc_len.weight"):
                return (0, 0 if n == "fc.weight" else 1)        # adjacent: [fc.weight ; fc_len.weight] = [K+1, H]
            if n in ("fc.bias", "fc_len.bias"):
                return (2, 0 if n == "fc.bias" else 1)          # adjacent: [fc.bias ; fc_len.bias] = [K+1]
            if n == "pos_embedding":
                return (2, 9)          # last of the small bucket: only its first S rows ever get a gradient (:190), so a
                                       # data-parallel step all-reduces the bucket up to row S and skips ~1 MB of zeros
            return (1, 0) if p.numel() % 4 == 0 else (2, 2)
        order = sorted(range(len(live)), key=lambda i: (key(live[i]), i))
        live = [live[i] for i in order]
        self.offsets, off = {}, 0
        for n, p in live:
            if n in ("depth_projection.weight", "pos_embedding"):
                off = (off + 3) // 4 * 4
            self.offsets[n] = (off, p.numel(), tuple(p.shape))
            off += p.numel()
        self.n_live = (off + 3) // 4 * 4
        off = self.n_live
        for n, p in dead:
            off = (off + 3) // 4 * 4
            self.offsets[n] = (off, p.numel(), tuple(p.shape))
            off += p.numel()
        self.n_total = (off + 3) // 4 * 4
        big0 = self.offsets["depth_projection.weight"][0] if "depth_projection.weight" in self.offsets else self.n_live
        self.bucket_small = (0, big0)           # (models without a depth projection: everything is the small bucket)
        self.bucket_big = (big0, self.n_live)
        self.params = torch.zeros(self.n_total, dtype=torch.float32, device=device)
        self.grads = torch.zeros(self.n_live, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(self.n_live, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(self.n_live, dtype=torch.float32, device=device)
        self.live_names = [n for n, _ in live]
        with torch.no_grad():
            for n, p in named:
                o, k, shp = self.offsets[n]
                self.params[o:o + k].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
                p.data = self.params[o:o + k].view(shp)
                p.grad = None

    def p(self, name):
        o, k, shp = self.offsets[name]
        return self.params[o:o + k].view(shp)

    def g(self, name):
        o, k, shp = self.offsets[name]
        return self.grads[o:o + k].view(shp)

    def attach_grads(self, named_params):
        """Expose the arena gradients as p.grad views (after a fused backward) so that any torch optimiser, gradient
        clipping or inspection code sees them.  Parameters the reference leaves without a gradient keep grad=None."""
        for n, p in named_params:
            p.grad = self.g(n) if self.is_live(n) else None


class _Shape:
    """Activation / gradient workspace for one (B, S, mode) shape."""

    def __init__(self, eng, B, S, train):
        dev, H, Q, K, L, heads = eng.device, eng.H, eng.Q, eng.K, eng.L, eng.heads
        N, BQ = B * S, B * Q
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)     # noqa: E731
        self.B, self.S, self.N, self.BQ = B, S, N, BQ
        self.rgb, self.dep, self.dep_pre = f(N, H), f(N, H), f(N, H)
        self.mean_d, self.rstd_d = f(N), f(N)
        if eng.bn:
            self.bn_mean, self.bn_rstd, self.bn_absg = f(2, H), f(2, H), f(2, H)
            self.bn_idx = torch.empty(2, max(1, int(H * 0.1)), dtype=torch.int64, device=dev)     # k = int(0.1 C) (:58)
        self.sums = torch.empty(2, H, dtype=torch.float64, device=dev)
        self.idx = torch.empty(2, H // 4, dtype=torch.int64, device=dev)
        self.mask = f(2, H)
        self.x0, self.h1, self.vsw, self.x1, self.h2 = f(2 * N, H), f(2 * N, H), f(2 * N, H), f(2 * N, H), f(2 * N, H)
        self.u, self.f1 = f(2 * N, 4 * H), f(2 * N, 4 * H)
        self.x3, self.y = f(2 * N, H), f(2 * N, H)
        self.m1, self.r1, self.m2, self.r2, self.mf, self.rf = f(2 * N), f(2 * N), f(2 * N), f(2 * N), f(2 * N), f(2 * N)
        self.fused = f(N, H)
        self.tgt0 = torch.zeros(BQ, H, dtype=torch.float32, device=dev)
        self.layers = []
        for _ in range(L):
            d = dict(sa_qkv=f(BQ, 3 * H), sa_o=f(BQ, H), p_sa=f(B, heads, Q, Q), t1_pre=f(BQ, H),
                     t1=f(BQ, H), m1=f(BQ), r1=f(BQ), caq=f(BQ, H), cakv=f(N, 2 * H), ca_o=f(BQ, H),
                     p_ca=f(B, heads, Q, S), t2_pre=f(BQ, H), t2=f(BQ, H), m2=f(BQ), r2=f(BQ), ff1=f(BQ, 4 * H),
                     t3_pre=f(BQ, H), t3=f(BQ, H), m3=f(BQ), r3=f(BQ))
            self.layers.append(d)
        self.tgtF, self.mF, self.rF = f(BQ, H), f(BQ), f(BQ)
        self.actdur = f(BQ, K + 1)
        self.seg = f(N, K)
        self.loss = f(4)
        self.tables = {}
        self.loss_ws = torch.zeros(ops.losses_ws_floats(B, S, eng.Q), dtype=torch.float32, device=dev)
        self.counts = torch.zeros(4, dtype=torch.int64, device=dev)
        if train:
            self.d_actdur, self.d_seg = torch.zeros(BQ, K + 1, dtype=torch.float32, device=dev), f(N, K)
            self.d_tgtF, self.d_t = f(BQ, H), f(BQ, H)
            # one buffer per gradient tensor: nothing is reused inside a step, so weight-gradient kernels on the side
            # stream can keep reading a tensor while the main stream moves on
            self.glayers = [dict(t3pre=f(BQ, H), ff2=f(BQ, H), ff1=f(BQ, 4 * H), t2=f(BQ, H), t2pre=f(BQ, H), cap=f(BQ, H),
                                 cao=f(BQ, H), caq=f(BQ, H), cakv=f(N, 2 * H), caqin=f(BQ, H), t1pre=f(BQ, H), sap=f(BQ, H),
                                 sao=f(BQ, H), saqkv=f(BQ, 3 * H), sain=f(BQ, H), t2b=f(BQ, H)) for _ in range(L)]
            self.d_fused, self.d_fused2 = f(N, H), f(N, H)
            lnw = lambda rows: f(max(ops.layernorm_bwd_ws_floats(rows, H), 4))       # noqa: E731
            self.lnp = dict(final=lnw(BQ), nf=lnw(2 * N), n2=lnw(2 * N), n1=lnw(2 * N), dep=lnw(N))
            # the fused embedding seam (embed.hip) leaves one (dgamma, dbeta) partial per frame for these two sites
            self.lnp_seam = dict(n1=f(2 * N * H), dep=f(2 * N * H))
            for l in range(L):
                self.lnp.update({f"d3_{l}": lnw(BQ), f"d2_{l}": lnw(BQ), f"d1_{l}": lnw(BQ)})
            self.d_x3, self.d_u, self.d_h1, self.d_h2, self.d_x1, self.d_v, self.d_x0 = (
                f(2 * N, H), f(2 * N, 4 * H), f(2 * N, H), f(2 * N, H), f(2 * N, H), f(2 * N, H), f(2 * N, H))
            self.d_rgb_pre, self.d_dep, self.d_dep_pre = f(N, H), f(N, H), f(N, H)
            self.d_h2b = f(2 * N, H)                     # second half-K partial of d_h2 (engine.split_k4h)
            if eng.bn:                                   # per-element terms of the BatchNorm / alpha parameter gradients
                self.bn_terms = [f(N, H) for _ in range(5)]
            # dropout keep-masks (one Philox launch fills the whole pool)
            sizes = dict(x0=2 * N * H)
            for l in range(L):
                sizes.update({f"sa_p{l}": B * heads * Q * Q, f"ca_p{l}": B * heads * Q * S, f"d1_{l}": BQ * H,
                              f"d2_{l}": BQ * H, f"d3_{l}": BQ * H, f"ff_{l}": BQ * 4 * H})
            tot = sum((v + 15) // 16 * 16 for v in sizes.values())
            self.drop_pool = torch.ones(tot, dtype=torch.uint8, device=dev)
            self.drop, o = {}, 0
            for k, v in sizes.items():
                self.drop[k] = self.drop_pool[o:o + v]
                o += (v + 15) // 16 * 16


class FusionEngine:
    def __init__(self, module, device):
        self.module = module
        self.device = torch.device(device)
        assert self.device.type == "cuda", "the HIP engine needs an MI355X device (there is no CPU path)"
        ops._lib.load()
        self.H, self.Q, self.K = module.hidden_dim, module.n_query, module.n_class
        self.heads, self.L = module.n_head, module.num_decoder_layers
        self.dh = self.H // self.heads
        self.pad_idx = module.src_pad_idx
        self.P, self.D = module.depth_projection.in_features, module.input_embed.in_features
        assert self.H % 8 == 0 and self.H % self.heads == 0
        # BN-blend fuser variant (model/futr_safuser_batchnormalization.py): BatchNorm on both embeddings, |gamma| scores,
        # alpha blend, no x_res; it takes the composed (un-paired, un-seamed) path around its own seam kernels
        self.bn = hasattr(module.fuser, "bn_rgb")
        self.arena = ParamArena(list(module.named_parameters()), self.device, BN_LIVE_PREFIXES if self.bn else ())
        self.ws = ops.GemmWorkspace(self.device)
        self._sel = {}                                  # constant 0/1 selection matrices of _wgrad_with_sums
        self.ws_side = ops.GemmWorkspace(self.device)
        self.side = torch.cuda.Stream(self.device)      # weight-gradient stream: off the backward's critical path
        self.side2 = torch.cuda.Stream(self.device)     # independent branch (self-attention of the queries)
        self.ws_side2 = ops.GemmWorkspace(self.device)
        # hidden >= 512 on one rank: the branch is 45-50 us long per direction there and pays for its join -- cfg4's per-GPU
        # shape 1.10 -> 1.08 ms, cfg5's 1.57 -> 1.44 (auto_side_stream; multi-rank flows keep one stream: their capture
        # with RCCL on the launch stream was never rehearsed with a second one).  The same switch forks the parameter-gradient
        # tail of the backward (overlap_param_tail below: the 108 us grouped weight-gradient launch beside the 270 us depth
        # weight gradient + its AdamW epilogue): cfg4 1.09 -> 1.03 ms
        self.auto_side_stream = True
        self.use_side_stream = False            # parameter-only branches on a second HIP stream: the cross-queue joins cost
                                                # more than the branches hide (352 vs 339 us/step at the bench shape)
        self.use_fused_decoder = False          # decoder.hip (one workgroup per clip and layer): measured 2 % slower than
                                                # the composed launches at the bench shape since the GEMM epilogue rework
        self.use_fused_tail = True              # last norm3 + decoder.norm + heads (and their adjoints): one launch each
        # training flows that call forward -> losses -> backward back to back set this: the tail's forward, the losses and
        # the tail's backward then run as ONE launch inside losses() (r3d_decoder_tail_losses); the anticipation outputs
        # of forward() are valid only after losses() in that mode
        self.defer_tail = False
        self.use_paired_launches = True          # one-layer decoder: independent GEMMs of the two chains share launches
        # hidden = 128: nn.Linear -> dropout -> residuals -> LayerNorm sites run as ONE row-complete launch (gemm_ln.hip)
        # instead of GEMM + LayerNorm: a dependent launch and a memory round trip less per site
        self.use_gemm_ln = True
        # hidden = 128, one decoder layer: the whole row-local neighbourhood of the SA-Fuser block (V projection + pair swap,
        # attn.proj, norm2, MLP, fuser.norm, pair mean, segmentation head, key/value projection) and the layer-0 query
        # self-attention sub-layer as ONE launch per direction (csrc/fuser_chain.hip) instead of 5 + 8
        self.use_fuser_chain = True
        # ... and the decoder layer's query side (cross-attention core, out_proj + norm2, FFN; in the training step also the
        # tail, the losses and the way back to the cross-attention's input gradients) as ONE launch (csrc/decoder_chain.hip)
        # instead of 4 + 1 + 5
        self.use_decoder_chain = True
        # the chain kernels' products on the bf16 matrix cores: every chain weight (and its transpose) is kept as three bf16
        # planes in MFMA operand order, rebuilt by one launch per step (ops.WeightPlanes); same exact three-way split and six
        # products as depth_prec = 1.  False: the exact-fp32 MFMA chain kernels
        self.chain_bf3 = True
        self._planes = None
        # Parallel branches of the step's hipGraph on the weight-gradient stream (self.side), both bit-neutral:
        #   overlap_planes: the re-split of the chain weights (they changed in the previous step's AdamW) runs beside the two
        #     input projections and the embedding seam instead of in front of the fuser chain;
        #   overlap_param_tail: everything of the backward that only feeds parameter gradients (the grouped weight-gradient
        #     launch with its folded sums) and the AdamW of every parameter but depth_projection.weight run beside the depth
        #     weight gradient (196 workgroups: a quarter of the chip idles under it) and the AdamW of that one weight (86 %
        #     of the model).  Needs gradients that are final when they are written (one rank, no exchange), and is joined at
        #     the end of backward() unless the caller promises the AdamW next (train_step).
        # MEASURED SLOWER, so both are off (kept under test and behind bench.py's --overlap-* flags): 0.1919 ms/step without,
        # 0.2063 with the planes branch, 0.2036 with the parameter branch, 0.2039 with both -- a fork/join pair between two
        # queues of a hipGraph costs 12-14 us here, more than the 7 / 14 us the branches hide (the same finding as round 1's
        # side-stream query branch; only the 0.5 ms Jacobi sweep is long enough to pay for its join).
        self.overlap_planes = False
        self.overlap_param_tail = False
        # ... what does pay: the re-split as RIDER workgroups of the embedding seam's launch (csrc/embed.hip: the seam's 128
        # workgroups are latency-bound and leave half the chip idle) -- one launch and ~6 us less per step
        self.ride_planes = True
        self._planes_rode = False
        self.fuse_depth_adamw = True             # see depth_adamw_fusable(); False: always the flat AdamW launch
        # the RGB embedding (input_embed, K = 2048) as a second product of the depth projection's launch: 61 K-splits x 4 tiles
        # occupy 244 of the 256 placed workgroups, the RGB product's 3 K-splits x 4 tiles take the other 12 -- one launch less,
        # no workgroup works longer than before; the RGB product then runs on the bf16 matrix cores too (same exact split)
        self.pair_embeddings = True
        self._planes_forked = False
        self._tail_pending = False
        self._overlap_tail_now = False
        # fused training flows (forward -> losses(tick=True) -> backward -> adamw(ticked=True)) may set this: the loss
        # kernel then leaves the reduction of its per-unit partials (loss / counter statistics only the host reads) to
        # one extra workgroup of the AdamW launch -- w.loss / w.counts are valid after adamw(), not after losses()
        self.defer_loss_reduce = False
        self.loss_acc = None                    # (float64[4], int64[4]) running sums the deferred reduction also feeds
        # the bias / broadcast-parameter / LayerNorm-parameter sums at the end of the backward as problems of the grouped
        # weight-gradient launch (a column sum over the rows of a residue class is a TN product with a constant 0/1
        # selection matrix) instead of a launch of their own
        self.fold_rowsums = True
        # the parameter-only query self-attention core of decoder layer 0 as extra workgroups of the gemm_ln launch
        self.ride_attention = True
        # ... and its backward core as extra workgroups of the fuser's norm2-backward launch (r3d_layernorm_bwd_multi_mha),
        # with the branch's input-projection gradient moved to the chain's last group: one launch fewer but measured SLOWER
        # (0.2382 -> 0.2410 ms/step: the K = 3H product lengthens the last group and the rider outlasts its host), so off
        self.ride_attention_bwd = False
        # the two K = 4H input-gradient products of the backward whose consumer is a LayerNorm backward (it adds two upstream
        # gradients anyway: dy + dy2) as TWO K = 2H problems of one grouped launch: measured neutral (0.2372 / 0.2391 against
        # 0.2391 / 0.2380 ms/step), so off; kept under test
        self.split_k4h = False
        # the two depth-projection GEMMs (83 % of the step's FLOPs) on the bf16 matrix cores through an exact three-way
        # operand split (csrc/gemm_bf3.hip; error per product <= 3 * 2^-24); 0 = the fp32 MFMA everywhere
        self.depth_prec = 1
        # (the fuser MLP's two forward products at hidden >= 512 through the same split-K bf16x3 kernel + reducer: measured
        #  SLOWER -- cfg4's per-GPU shape 1.157 -> 1.236 ms, cfg5's 1.716 -> 1.746: two K-splits of 8 k-steps pay the kernel's
        #  prologue / epilogue twice and the reducer re-reads 16 MB of slabs; they stay on the fp32 64 x 64 tiles)
        self.use_fused_embed = not self.bn       # train mode: projections' slab sums + LN + exchange + norm1 in one launch
        # False: train()-state steps without dropout (parity runs against a reference whose dropout probabilities were
        # set to 0; RNG streams cannot match).  Read from the module so that it survives model.to() re-creating the engine.
        self.dropout_enabled = bool(getattr(module, "r3d_dropout_enabled", True))
        self.erank_weight = 0.0           # > 0: total loss -= erank_weight * effective_rank(fused token matrix) (build-side
                                          # rank-enhancing penalty; the reference only describes it, README.md:8-14)
        # Jacobi on X V0 (V0 = the previous step's right singular basis, carried through the sweep): pays when the tokens
        # move little between steps (late training: 3-5 sweeps instead of 9-11, each 1.7x dearer with V riding along);
        # at lr 1e-3 from a random init the tokens move 20-100 % per step and no basis survives (measured: 1.54 vs
        # 1.27 ms/step at the headline shape), so it is opt-in
        self.erank_warm_start = False
        self.erank_max_sweeps = 16        # blocked (two-level) Jacobi: sweeps enqueued per step (8-10 are used; the rest no-op)
        # the Jacobi forward depends on the fused tokens only: it is enqueued on a second stream as soon as they exist (a
        # parallel branch of the step's hipGraph) and joined where its backward adds into the fuser's upstream gradient --
        # the decoder's forward, the losses and the decoder's backward run beside it
        self.erank_side_stream = True
        self.er_stream = torch.cuda.Stream(self.device)
        self.ws_er = ops.GemmWorkspace(self.device)
        self._er_pending = False
        self.shapes = {}
        self.train_mask = None            # cached train-mode selection (data independent)
        self.drop_seed = 0x5EED
        self.drop_offset = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.step_t = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._lr_host = None
        self.dur_den = None               # device scalar set by the data-parallel wrapper
        self.score_allreduce = None       # callable(sums fp64 [2,H]) -> global row count, set by the DP wrapper
        self.bn_sync = None               # parallel.SyncBatchNorm, set by the DP wrapper for the BN-blend variant
        self.grad_hook = None             # callable(stage) set by the DP wrapper: "small_ready" / "big_ready"
        self.tp = None                    # parallel.PixelShardedDepth: depth_projection tensor-parallel over pixels
        self._fw = None
        self.last = None
        self._adam = None
        self._drop_ready = None           # workspace whose dropout pool already holds the masks of the next forward
        a = self.arena
        K, H = self.K, self.H
        o_w = a.offsets["fc.weight"][0]
        self.w_head = a.params[o_w:o_w + (K + 1) * H].view(K + 1, H)
        self.gw_head = a.grads[o_w:o_w + (K + 1) * H].view(K + 1, H)
        o_b = a.offsets["fc.bias"][0]
        assert a.offsets["fc_len.weight"][0] == o_w + K * H and a.offsets["fc_len.bias"][0] == o_b + K
        self.b_head = a.params[o_b:o_b + K + 1]
        self.gb_head = a.grads[o_b:o_b + K + 1]

    # ------------------------------------------------------------------------------------------------------
    def _shape(self, B, S, train):
        key = (B, S, bool(train))
        if key not in self.shapes:
            self.shapes[key] = _Shape(self, B, S, train)
        return self.shapes[key]

    def _train_masks(self, B, S):
        """train mode: score = |d(mean)/dx| averaged over (B,T) = the constant 1/(B*T*C) for every channel
        (futr_safuser_tokenfusion.py:40-45) -> selection is pure tie-breaking and depends on C only."""
        if self.train_mask is None:
            C = self.H
            sc = torch.full((2, C), 1.0 / (B * S * C), dtype=torch.float32, device=self.device)
            idx = torch.empty(2, C // 4, dtype=torch.int64, device=self.device)
            mask = torch.empty(2, C, dtype=torch.float32, device=self.device)
            ops.token_select(C // 4, idx, mask, score_f=sc)
            self.train_mask = (idx, mask)
        return self.train_mask

    # ------------------------------------------------------------------------------------------------------
    def forward(self, feats, depth, labels, mode="train", training=False, need_grad=True, bn_training=None):
        """feats [B,S,D] f32, depth [B,S,...] f32 (flattened to [N,P]), labels [B,S] int64 (train mode only).
        bn_training (BN-blend variant): batch statistics + running-stat update; default = `training`.
        Returns dict of views into the workspace: seg [B,S,K], action [B,Q,K], duration [B,Q] (strided views)."""
        self.forward_begin(feats, depth, labels, mode, training, need_grad)
        self._fw["bn_training"] = training if bn_training is None else bn_training
        if self._fw["tp"] is not None:
            self._fw["tp"].exchange_forward(self._fw["w"])
        return self.forward_finish()

    def forward_begin(self, feats, depth, labels, mode="train", training=False, need_grad=True):
        """The two input projections (:179,194-195).  With a pixel-sharded depth projection this rank multiplies ITS
        pixel columns of every rank's clips and the partial sums are exchanged before forward_finish()."""
        a, H = self.arena, self.H
        B, S = feats.shape[0], feats.shape[1]
        N = B * S
        assert feats.is_cuda and depth.is_cuda and feats.dtype == torch.float32 and depth.dtype == torch.float32
        x_rgb = feats.reshape(N, -1)
        x_dep = depth.reshape(N, -1)
        assert x_rgb.shape[1] == self.D and x_dep.shape[1] == self.P, (x_rgb.shape, x_dep.shape, self.D, self.P)
        assert x_rgb.is_contiguous() and x_dep.is_contiguous()
        w = self._shape(B, S, need_grad)
        self._loss_pending = None          # (see losses(): a pending loss reduction never survives into another step)
        self._erank_join()                 # (a forward whose backward never ran: its sweep still reads the workspace)
        drop = training and need_grad and self.dropout_enabled
        if drop:
            if self._drop_ready is w:     # the previous step's AdamW launch already filled the pool for this offset
                self._drop_ready = None
            else:
                ops.dropout_mask(w.drop_pool, DROP_P, self.drop_seed, self.drop_offset)
        tp = self.tp if (self.tp is not None and need_grad) else None
        seam = self.use_fused_embed and mode == "train" and H <= 1024
        self._join_tail()
        if (self.overlap_planes and self.chain_bf3 and self.L == 1 and (self.use_fuser_chain or self.use_decoder_chain) and
                self._chain_shape_ok(w)):
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.chain_planes().refresh()
            self._planes_forked = True
        # (running this GEMM on the side stream beside the 5x longer depth projection was measured: the cross-queue
        #  join costs more than the 5 us it hides)
        rgb_done = None
        dr = d = None
        if seam and tp is None and self.depth_prec == 1 and self.pair_embeddings:
            # both input projections in ONE launch: the RGB embedding's K-splits take the workgroup slots the depth
            # projection's leave empty (ops.gemm_bf3_nt_pair); bias / ReLU / LayerNorm are the seam's either way
            pr = ops.gemm_bf3_nt_pair(x_dep, a.p("depth_projection.weight"), w.dep_pre, self.ws,
                                      x_rgb, a.p("input_embed.weight"), w.rgb, self.ws_side)
            if pr is not None:
                d, dr = pr
        if dr is None:
            # (prec: the planner takes the bf16x3 split-K kernel for it only where the output is large -- hidden >= 256: 29.9 ->
            #  ~15 us at cfg4's per-GPU shape; the headline shape's RGB product rides in the pair launch above)
            dr = ops.gemm(GEMM_NT, x_rgb, a.p("input_embed.weight"), w.rgb, bias=a.p("input_embed.bias"), act=1,
                          ws=self.ws_side if seam else self.ws, defer_reduce=seam, prec=self.depth_prec)
        slabs_r = self.ws_side.buf if (seam and dr.splitk > 1) else None
        if d is not None:
            pass
        elif tp is None:
            d = ops.gemm(GEMM_NT, x_dep, a.p("depth_projection.weight"), w.dep_pre, bias=a.p("depth_projection.bias"),
                         ws=self.ws, defer_reduce=True, prec=self.depth_prec)    # (:194-195)
        else:
            tp.partial_forward(w, x_dep, self.ws)
        # the split-K slabs of the depth GEMM are consumed by the LayerNorm launch in forward_finish(): keep THIS buffer
        # (a later, larger request would re-allocate the workspace) -- nothing may use self.ws before that launch
        slabs = self.ws.buf if (d is not None and d.splitk > 1) else None
        self._fw = dict(w=w, x_rgb=x_rgb, x_dep=x_dep, labels=labels, mode=mode, drop=drop, d=d, tp=tp, B=B, S=S,
                        slabs=slabs, seam=seam, dr=dr, slabs_r=slabs_r, rgb_done=rgb_done)

    def forward_finish(self):
        """Everything after the input projections; see forward()."""
        fw = self._fw
        w, x_rgb, x_dep, labels, mode, drop, d, tp = (fw[k] for k in ("w", "x_rgb", "x_dep", "labels", "mode", "drop", "d",
                                                                       "tp"))
        B, S = fw["B"], fw["S"]
        a, H, Q, K, heads, dh = self.arena, self.H, self.Q, self.K, self.heads, self.dh
        N, BQ = B * S, B * Q
        dsc = 1.0 / (1.0 - DROP_P)
        dm = (lambda k: w.drop[k]) if drop else (lambda k: None)
        key_labels = None
        if mode == "train":                     # get_pad_mask (:168,243-244) is evaluated inside the attention kernel
            assert labels.dtype == torch.int64 and labels.is_cuda and labels.is_contiguous()
            key_labels = labels
        main = torch.cuda.current_stream()
        multi = self._multi_stream()
        s2 = self.side2 if multi else main
        ws2 = self.ws_side2 if multi else self.ws
        qpos = a.p("query_embed.weight")
        pos = a.p("pos_embedding")[0, :S]

        def sa_block(l, tgt_in, wsx):
            """self-attention sub-layer of decoder layer l + the query projection of its cross-attention
            (transformer.py:289-293,300); for layer 0 (tgt = 0) it depends on parameters only."""
            c, pl = w.layers[l], f"transformer.decoder.layers.{l}."
            ops.gemm(GEMM_NT, tgt_in, a.p(pl + "self_attn.in_proj_weight"), c["sa_qkv"], a_add=qpos, a_add_mod=Q,
                     bias=a.p(pl + "self_attn.in_proj_bias"), ws=wsx)
            ops.mha_core_fwd(c["sa_qkv"][:, :H], c["sa_qkv"][:, H:2 * H], c["sa_qkv"][:, 2 * H:], c["p_sa"], c["sa_o"], B,
                             heads, Q, Q, dh, drop_mask=dm(f"sa_p{l}"), drop_scale=dsc)
            if self._gln(BQ, H):
                ops.gemm_ln_fwd([dict(a=c["sa_o"], w=a.p(pl + "self_attn.out_proj.weight"),
                                      bias=a.p(pl + "self_attn.out_proj.bias"), drop_mask=self._dm2(dm(f"d1_{l}"), BQ, H),
                                      drop_scale=dsc, res1=None if l == 0 else tgt_in, pre=c["t1_pre"],
                                      gamma=a.p(pl + "norm1.weight"), beta=a.p(pl + "norm1.bias"), y=c["t1"], mean=c["m1"],
                                      rstd=c["r1"])])
            else:
                ops.gemm(GEMM_NT, c["sa_o"], a.p(pl + "self_attn.out_proj.weight"), c["t1_pre"],
                         bias=a.p(pl + "self_attn.out_proj.bias"), drop_mask=self._dm2(dm(f"d1_{l}"), BQ, H),
                         drop_scale=dsc, res1=None if l == 0 else tgt_in, ws=wsx)                 # layer 0: tgt = 0 (:209)
                ops.layernorm_fwd(c["t1_pre"], a.p(pl + "norm1.weight"), a.p(pl + "norm1.bias"), c["t1"], c["m1"], c["r1"])
            wi, bi = a.p(pl + "multihead_attn.in_proj_weight"), a.p(pl + "multihead_attn.in_proj_bias")
            ops.gemm(GEMM_NT, c["t1"], wi[:H], c["caq"], a_add=qpos, a_add_mod=Q, bias=bi[:H], ws=wsx)

        seam = fw["seam"]
        pre = "fuser.blocks.0."
        if fw["rgb_done"] is not None:
            main.wait_event(fw["rgb_done"])
        if seam:
            # ---- one launch: slab sums of both projections, bias/ReLU, depth LayerNorm + ReLU (:183,195-197), token
            # exchange + embd_drop (:56-62,83) and the fuser block's norm1 (transformerblock.py:122)
            idx, mask = self._train_masks(B, S)
            dr = fw["dr"]
            if tp is not None:
                dep_src, ns_d, bias_d = tp.summed(w), 1, a.p("depth_projection.bias")
            elif d.splitk > 1:
                dep_src, ns_d, bias_d = fw["slabs"], d.splitk, a.p("depth_projection.bias")
            else:
                dep_src, ns_d, bias_d = w.dep_pre, 1, None
            ride = bool(self.ride_planes and self.chain_bf3 and not self._planes_forked and self.L == 1 and H < 512 and
                        not self._multi_stream() and not self.use_fused_decoder and self.use_paired_launches and
                        ((self.use_fuser_chain and self._chain_shape_ok(w)) or self._dec_chain_ok(w)))
            ops.embed_fuse_fwd(fw["slabs_r"] if dr.splitk > 1 else w.rgb, dr.splitk if dr.splitk > 1 else 0,
                               a.p("input_embed.bias"), dep_src, ns_d, bias_d, a.p("depth_layernorm.weight"),
                               a.p("depth_layernorm.bias"), mask[0], mask[1], dm("x0"), dsc, a.p(pre + "norm1.weight"),
                               a.p(pre + "norm1.bias"), w.rgb, w.dep_pre, w.mean_d, w.rstd_d, w.dep, w.x0, w.h1, w.m1, w.r1,
                               planes=self.chain_planes() if ride else None)
            self._planes_rode = ride
        # ---- depth LayerNorm + ReLU first: it drains the deferred split-K slabs (:196-197)
        elif tp is not None:                      # the exchanged sum of the ranks' partial products, bias still to add
            ops.layernorm_fwd(tp.summed(w), a.p("depth_layernorm.weight"), a.p("depth_layernorm.bias"), w.dep, w.mean_d,
                              w.rstd_d, relu=True, nsplit=1, bias=a.p("depth_projection.bias"), pre_out=w.dep_pre,
                              rows=N, H=H)
        elif d.splitk > 1:
            ops.layernorm_fwd(fw["slabs"], a.p("depth_layernorm.weight"), a.p("depth_layernorm.bias"), w.dep, w.mean_d,
                              w.rstd_d, relu=True, nsplit=d.splitk, bias=a.p("depth_projection.bias"),
                              pre_out=w.dep_pre, rows=N, H=H)                    # (:196-197)
        else:
            ops.layernorm_fwd(w.dep_pre, a.p("depth_layernorm.weight"), a.p("depth_layernorm.bias"), w.dep, w.mean_d,
                              w.rstd_d, relu=True)
        fused_dec = self.use_fused_decoder and ops.decoder_fused_supported(S, Q, H, heads)
        if fused_dec:
            multi = False                     # the whole layer is one launch: nothing left to branch
        # One decoder layer on one stream: the query self-attention sub-layer depends on parameters only, the fuser block
        # only on the embeddings -- their GEMMs are paired into shared launches (ops.GemmGroup) instead of queueing
        # behind each other.  Otherwise: branch s2 (or inline) first, then the fuser chain.
        # (hidden >= 512: the shared launches' single tile and missing split-K cost more than the launches they save --
        #  cfg5's per-GPU shape 1.87 -> 1.69 ms/step unpaired, cfg4's 1.191 -> 1.179)
        paired = self.use_paired_launches and (not fused_dec) and (not multi) and self.L == 1 and H < 512
        if multi and fw["rgb_done"] is None:
            s2.wait_stream(main)              # (otherwise forward_begin already forked this branch)
        if not fused_dec and not paired:
            with torch.cuda.stream(s2):
                sa_block(0, w.tgt0, ws2)
        # ---- token selection + exchange (:33-66)
        if seam or self.bn:
            pass
        elif mode == "train":
            idx, mask = self._train_masks(B, S)
        else:
            ops.colabssum(w.rgb, w.sums[0])
            ops.colabssum(w.dep, w.sums[1])
            count = float(N)
            if self.score_allreduce is not None:
                count = self.score_allreduce(w.sums, N)
            ops.token_select(H // 4, w.idx, w.mask, score_sum=w.sums, count=count)
            idx, mask = w.idx, w.mask
        # ---- SA-Fuser block in closed form (transformerblock.py:118-135) + x_res + norm + mean (:86-94)
        if self.bn:
            # BatchNorm statistics (+ running-stat update), |gamma| scores, k = int(0.1 C) smallest, alpha blend + dropout
            # + norm1 (futr_safuser_batchnormalization.py:45-75,95; transformerblock.py:122)
            mod = self.module.fuser
            bt = bool(fw.get("bn_training", False))
            if bt and self.bn_sync is not None:     # data parallel: statistics over the global batch (parallel.SyncBatchNorm)
                self.bn_sync.forward(self, w, mod)
            else:
                ops.bn_stats(w.rgb, w.dep, mod.bn_rgb, mod.bn_depth, w.bn_mean, w.bn_rstd, w.bn_absg, bt)
            ops.token_select(w.bn_idx.shape[1], w.bn_idx, w.mask, score_f=w.bn_absg)
            idx, mask = w.bn_idx, w.mask
            ops.bn_blend_fwd(w.rgb, w.dep, w.bn_mean, w.bn_rstd, a.p("fuser.bn_rgb.weight"), a.p("fuser.bn_rgb.bias"),
                             a.p("fuser.bn_depth.weight"), a.p("fuser.bn_depth.bias"), a.p("fuser.alpha").view(-1), mask[0],
                             mask[1], dm("x0"), dsc, a.p(pre + "norm1.weight"), a.p(pre + "norm1.bias"), w.x0, w.h1, w.m1,
                             w.r1)
        elif not seam:
            ops.token_exchange_fwd(w.rgb, w.dep, mask[0], mask[1], w.x0, drop_mask=dm("x0"), drop_scale=dsc)
            ops.layernorm_fwd(w.x0, a.p(pre + "norm1.weight"), a.p(pre + "norm1.bias"), w.h1, w.m1, w.r1)
        wv = a.p(pre + "attn.qkv.weight")[2 * H:]
        if paired:
            self._forward_paired(w, fw, dm, dsc, drop, wv, pre, qpos, pos)
        else:
            ops.gemm(GEMM_NT, w.h1, wv, w.vsw, c_row_xor=1, ws=self.ws)        # V of the OTHER modality token
            gln = self._gln(2 * N, H) and self._gln(2 * N, 4 * H)
            if gln:
                ops.gemm_ln_fwd([dict(a=w.vsw, w=a.p(pre + "attn.proj.weight"), bias=a.p(pre + "attn.proj.bias"),
                                      res1=w.x0, pre=w.x1, gamma=a.p(pre + "norm2.weight"), beta=a.p(pre + "norm2.bias"),
                                      y=w.h2, mean=w.m2, rstd=w.r2)])
            else:
                ops.gemm(GEMM_NT, w.vsw, a.p(pre + "attn.proj.weight"), w.x1, bias=a.p(pre + "attn.proj.bias"), res1=w.x0,
                         ws=self.ws)
                ops.layernorm_fwd(w.x1, a.p(pre + "norm2.weight"), a.p(pre + "norm2.bias"), w.h2, w.m2, w.r2)
            ops.gemm(GEMM_NT, w.h2, a.p(pre + "mlp.mlp.0.weight"), w.f1, bias=a.p(pre + "mlp.mlp.0.bias"), act=2,
                     pre_out=w.u, ws=self.ws)
            if gln:
                ops.gemm_ln_fwd([dict(a=w.f1, w=a.p(pre + "mlp.mlp.2.weight"), bias=a.p(pre + "mlp.mlp.2.bias"), res1=w.x1,
                                      res2=None if self.bn else w.x0, pre=w.x3, gamma=a.p("fuser.norm.weight"),
                                      beta=a.p("fuser.norm.bias"), y=w.y, mean=w.mf, rstd=w.rf, pair_out=w.fused)])
            else:
                ops.gemm(GEMM_NT, w.f1, a.p(pre + "mlp.mlp.2.weight"), w.x3, bias=a.p(pre + "mlp.mlp.2.bias"), res1=w.x1,
                         res2=None if self.bn else w.x0, ws=self.ws)        # (the BN-blend variant has no x_res, :97,101)
                ops.layernorm_fwd(w.x3, a.p("fuser.norm.weight"), a.p("fuser.norm.bias"), w.y, w.mf, w.rf,
                                  pair_out=w.fused)
        if not paired:
            self._erank_fork(w)
        # ---- segmentation head (:228-232); with the composed decoder it shares a launch with the layer-0 key/value
        # projection (both read `fused`, neither depends on the other)
        if paired:
            pass                              # (in _forward_paired, together with the cross-attention query projection)
        elif not fused_dec:
            if not hasattr(w, "fwd_group"):
                wi0 = a.p("transformer.decoder.layers.0.multihead_attn.in_proj_weight")
                bi0 = a.p("transformer.decoder.layers.0.multihead_attn.in_proj_bias")
                w.fwd_group = ops.GemmGroup(GEMM_NT, [
                    dict(a=w.fused, b=a.p("fc_seg.weight"), c=w.seg, bias=a.p("fc_seg.bias")),
                    dict(a=w.fused, b=wi0[H:], c=w.layers[0]["cakv"], bias=bi0[H:], a_add=pos, a_add_mod=S)],
                    tile=1 if H < 256 else 2)
            w.fwd_group.launch()
        else:
            ops.gemm(GEMM_NT, w.fused, a.p("fc_seg.weight"), w.seg, bias=a.p("fc_seg.bias"), ws=self.ws)
        # ---- decoder (transformer.py:75-128,161-191,281-330); memory = fused, encoder bypassed (:77-78)
        self.last_drop_flag = bool(drop)
        if fused_dec:
            self._decoder_fused(w, key_labels, drop, dsc)
        else:
            self._decoder_unfused(w, key_labels, dm, dsc, multi, main, s2, sa_block, paired)
        if self._planes_forked:                # (no chain launch consumed the branch: eval shapes the decoder chain skips)
            main.wait_stream(self.side)
            self._planes_forked = False
        self._planes_rode = False
        er = self.erank_weight != 0.0 and hasattr(w, "glayers")
        if er and not w.__dict__.get("_er_forked", False):
            self._erank_forward(w)
        w._er_forked = False
        self.last = dict(w=w, x_rgb=x_rgb, x_dep=x_dep, mask=mask, idx=idx, drop=drop, mode=mode, tp=tp, seam=seam, erank=er,
                         paired=paired, bn_training=bool(fw.get("bn_training", False)))
        return dict(seg=w.seg.view(B, S, K), action=w.actdur[:, :K].view(B, Q, K), duration=w.actdur[:, K].view(B, Q))

    # ---- effective-rank penalty on the fused token matrix [N, H] (erank.hip; Appendix A.11) ------------------------------
    def _erank_fork(self, w):
        """Called where the fused tokens have just been enqueued: the Jacobi forward goes to the side stream."""
        if self.erank_weight == 0.0 or not hasattr(w, "glayers") or not self.erank_side_stream:
            return
        self.er_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.er_stream):
            self._erank_forward(w)
        w._er_forked = True
        self._er_pending = True

    def _erank_join(self):
        if self._er_pending:
            torch.cuda.current_stream().wait_stream(self.er_stream)
            self._er_pending = False

    def _erank_forward(self, w):
        """Jacobi forward on the fused tokens.  Warm start (erank_warm_start): the sweep runs on X V0, V0 the right
        singular basis the previous step's sweep left behind (the kernel applies its rotations to V too; identity on the
        first step) -- when the tokens move little between optimiser steps X V0 has almost orthogonal columns and 3-5
        sweeps replace 10-11.  The singular values of X V0 are those of X (V0 orthogonal), and the rotated columns
        Af = X V0 V' are the same (X V) the backward needs."""
        N, H = w.N, self.H
        if not hasattr(w, "er_sigma"):
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)     # noqa: E731
            w.er_gout = f(1)
            w.er_blk = None
            if ops.erank_fits(N, H):
                w.er_sigma, w.er_stats, w.er_af = f(1, H), f(1, 4), f(1, H, N)
                w.er_vt = torch.eye(H, dtype=torch.float32, device=self.device)      # V0^T
                w.er_vraw, w.er_vg, w.er_gv, w.er_xw = f(H, H), f(H, H), f(H, H), f(N, H)
            else:
                # beyond one CU's LDS (cfg4 / cfg5 shapes): the two-level block Jacobi with the columns in HBM, on the
                # orientation with the fewer columns (the singular values of X and X^T are the same) -- fused^T is
                # handed over as it lies in memory, no transposition pass
                w.er_flip = N < H
                R, Cc = (H, N) if w.er_flip else (N, H)
                w.er_blk = ops.ErankBlockedBufs(R, Cc, self.device, max_sweeps=self.erank_max_sweeps)
                w.er_sigma, w.er_stats = w.er_blk.sigma.view(1, Cc), w.er_blk.stats.view(1, 4)
        if w.er_blk is not None:
            ops.erank_blocked_into(w.fused, w.er_blk, transposed=w.er_flip)
        elif self.erank_warm_start and ops.erank_fits_warm(N, H):
            wsx = self.ws_er
            ops.gemm(GEMM_NT, w.fused, w.er_vt, w.er_xw, ws=wsx)            # X V0
            ops.erank_jacobi_warm(w.er_xw, w.er_sigma, w.er_stats, w.er_vraw, vt_in=w.er_vt, af_t=w.er_af)
            # the carried basis is a product of ever more rotations: one Newton-Schulz step per use keeps its departure
            # from orthogonality at rounding level (V^T <- 1.5 V^T - 0.5 (V^T V) V^T)
            ops.gemm(GEMM_NT, w.er_vraw, w.er_vraw, w.er_vg, ws=wsx)
            ops.gemm(GEMM_NN, w.er_vg, w.er_vraw, w.er_gv, ws=wsx)
            ops.erank_vt_polish(w.er_vraw, w.er_gv, w.er_vt)
        else:
            ops.erank_jacobi(w.fused, w.er_sigma, w.er_stats, af_t=w.er_af)

    def _erank_backward(self, w, ws, dst=None):
        """d_fused2 += d(-erank_weight * erank)/d(fused)  (dst given: dst = that gradient, written, not accumulated): U diag(g)
        V^T from the rotated columns the sweep left behind, in the Neumann-corrected form of r3d_amd.erank.ErankBackward;
        when the sweep ran on fused^T (wide token matrices) the same products in the transposed orientation."""
        from .erank import ErankBackward
        if not hasattr(w, "er_bwd"):
            w.er_bwd = ErankBackward(w.N, self.H, w.er_blk is not None and w.er_flip, self.device)
        w.er_gout.fill_(-float(self.erank_weight))
        A = w.er_blk.af_t if w.er_blk is not None else w.er_af[0]
        w.er_bwd.run(w.fused, A, w.er_sigma[0], w.er_stats[0], w.er_gout, dst if dst is not None else w.d_fused2,
                     dst is None, ws)

    def erank_value(self):
        """Effective rank of the last forward's fused tokens (device scalar; valid when erank_weight != 0)."""
        self._erank_join()
        return self.last["w"].er_stats[0, 0]

    def chain_planes(self):
        """The bf16x3 planes of the chain kernels' weights (built on first use; refresh() re-splits the current parameters)."""
        if self._planes is None:
            a, H = self.arena, self.H
            pre, pl = "fuser.blocks.0.", "transformer.decoder.layers.0."
            wi = a.p(pl + "multihead_attn.in_proj_weight")
            W = dict(wv=a.p(pre + "attn.qkv.weight")[2 * H:], wproj=a.p(pre + "attn.proj.weight"), w1=a.p(pre + "mlp.mlp.0.weight"),
                     w2=a.p(pre + "mlp.mlp.2.weight"), wkv=wi[H:], wseg=a.p("fc_seg.weight"),
                     dec_wo=a.p(pl + "multihead_attn.out_proj.weight"), dec_w1=a.p(pl + "linear1.weight"),
                     dec_w2=a.p(pl + "linear2.weight"))
            ent = {}
            for k, w_ in W.items():
                ent["pl_" + k] = (w_, False)             # B[n][k] = W[n][k]: the forward products y = x W^T
                ent["pl_" + k + "_t"] = (w_, True)       # B[n][k] = W[k][n]: the input-gradient products dx = dy W
            self._planes = ops.WeightPlanes(ent, self.device)
        return self._planes

    def _multi_stream(self):
        """The query self-attention branch (parameters only for a one-layer decoder) on a second stream?"""
        return bool(self.use_side_stream or (self.auto_side_stream and self.H >= 512 and self.tp is None and
                                             self.grad_hook is None))

    def _planes_refresh(self):
        """The planes of the current parameters, before the first chain launch of a forward: joins the branch forward_begin
        forked, or re-splits here."""
        if self._planes_forked:
            torch.cuda.current_stream().wait_stream(self.side)
            self._planes_forked = False
        elif self._planes_rode:                 # (the seam's launch of this forward carried the re-split)
            pass
        else:
            self.chain_planes().refresh()

    def _join_tail(self):
        """Joins the parameter-gradient branch of backward() (overlap_param_tail) into the current stream."""
        if self._tail_pending:
            torch.cuda.current_stream().wait_stream(self.side)
            self._tail_pending = False

    def _chain_shape_ok(self, w):
        return bool(not self.bn and self.dh == 16 and ops.fuser_chain_supported(w.N, self.H, self.K, w.B, self.Q, self.heads))

    def _chain_ok(self, w):
        return bool(self.use_fuser_chain and self._chain_shape_ok(w))

    def _dec_chain_ok(self, w):
        return bool(self.use_decoder_chain and self.L == 1 and self.dh == 16 and w.BQ <= 1024 and
                    ops.decoder_chain_supported(self.H, self.Q, self.heads, w.S))

    def _dec_chain(self, w, drop):
        """The argument block of the decoder chain kernel for this workspace (one per dropout state)."""
        bf3 = bool(self.chain_bf3)
        key = ("dec_chain", bool(drop), bf3)
        if bf3 and not (self.use_fuser_chain and self._chain_shape_ok(w)):
            self._planes_refresh()                       # (otherwise the fuser chain's forward launch refreshed them already)
        if key not in w.tables:
            pls = self.chain_planes() if bf3 else None
            dpl = None
            if pls is not None:
                dpl = dict(pl_wo=pls.ptr("pl_dec_wo"), pl_w1=pls.ptr("pl_dec_w1"), pl_w2=pls.ptr("pl_dec_w2"),
                           pl_w2_t=pls.ptr("pl_dec_w2_t"), pl_w1_t=pls.ptr("pl_dec_w1_t"), pl_wo_t=pls.ptr("pl_dec_wo_t"))
            a, H = self.arena, self.H
            c, pl = w.layers[0], "transformer.decoder.layers.0."
            dk = (lambda k: w.drop[k]) if drop else (lambda k: None)
            gl = w.glayers[0] if hasattr(w, "glayers") else None
            g = (lambda k: gl[k]) if gl is not None else (lambda k: None)
            w.tables[key] = ops.DecoderChain(
                caq=c["caq"], cakv=c["cakv"], p_ca=c["p_ca"], drop_ca=dk("ca_p0"), ca_o=c["ca_o"],
                wo=a.p(pl + "multihead_attn.out_proj.weight"), bo=a.p(pl + "multihead_attn.out_proj.bias"), drop_d2=dk("d2_0"),
                t1=c["t1"], t2_pre=c["t2_pre"], g2=a.p(pl + "norm2.weight"), be2=a.p(pl + "norm2.bias"), t2=c["t2"], m2=c["m2"],
                r2=c["r2"], w1=a.p(pl + "linear1.weight"), b1=a.p(pl + "linear1.bias"), drop_ff=dk("ff_0"), ff1=c["ff1"],
                w2=a.p(pl + "linear2.weight"), b2=a.p(pl + "linear2.bias"), drop_d3=dk("d3_0"), t3_pre=c["t3_pre"],
                d_t3pre=g("t3pre"), d_ff2=g("ff2"), d_ff1=g("ff1"), d_t2pre=g("t2pre"), d_cap=g("cap"), d_cao=g("cao"),
                d_caq=g("caq"), d_cakv=g("cakv"), part_d2=w.lnp["d2_0"] if gl is not None else None,
                drop_scale=1.0 / (1.0 - DROP_P), pad_idx=self.pad_idx, B=w.B, S=w.S, H=H, Q=self.Q, heads=self.heads, planes=dpl)
        return w.tables[key]

    def _gln(self, rows, K):
        return self.use_gemm_ln and ops.gemm_ln_supported(rows, K, self.H)

    def _forward_paired(self, w, fw, dm, dsc, drop, wv, pre, qpos, pos):
        """Fuser block (transformerblock.py:118-135, :86-94) interleaved with the layer-0 query self-attention sub-layer
        (transformer.py:289-293,300): independent GEMMs share a launch."""
        a, H, Q, S, B, BQ, heads, dh = self.arena, self.H, self.Q, w.S, w.B, w.BQ, self.heads, self.dh
        c, pl = w.layers[0], "transformer.decoder.layers.0."
        if self._chain_ok(w):
            bf3 = bool(self.chain_bf3)
            key = ("fwd_chain", bool(drop), bf3)
            if bf3:
                self._planes_refresh()                   # the parameters may have changed since the last forward: re-split
            if key not in w.tables:
                pls = self.chain_planes() if bf3 else None
                wi, bi = a.p(pl + "multihead_attn.in_proj_weight"), a.p(pl + "multihead_attn.in_proj_bias")
                w.tables[key] = ops.FuserChainFwd(
                    x0=w.x0, h1=w.h1, wv=wv, wproj=a.p(pre + "attn.proj.weight"), bproj=a.p(pre + "attn.proj.bias"),
                    g2=a.p(pre + "norm2.weight"), be2=a.p(pre + "norm2.bias"), w1=a.p(pre + "mlp.mlp.0.weight"),
                    b1=a.p(pre + "mlp.mlp.0.bias"), w2=a.p(pre + "mlp.mlp.2.weight"), b2=a.p(pre + "mlp.mlp.2.bias"),
                    gf=a.p("fuser.norm.weight"), bef=a.p("fuser.norm.bias"), pos=pos.contiguous(), wkv=wi[H:], bkv=bi[H:],
                    wseg=a.p("fc_seg.weight"), bseg=a.p("fc_seg.bias"), vsw=w.vsw, x1=w.x1, h2=w.h2, m2=w.m2, r2=w.r2, u=w.u,
                    f1=w.f1, x3=w.x3, y=w.y, mf=w.mf, rf=w.rf, fused=w.fused, seg=w.seg, cakv=c["cakv"],
                    qpos=qpos, w_in=a.p(pl + "self_attn.in_proj_weight"), b_in=a.p(pl + "self_attn.in_proj_bias"),
                    w_out=a.p(pl + "self_attn.out_proj.weight"), b_out=a.p(pl + "self_attn.out_proj.bias"),
                    g1=a.p(pl + "norm1.weight"), be1=a.p(pl + "norm1.bias"), wq=wi[:H], bq=bi[:H],
                    drop_sa=dm("sa_p0"), drop_d1=dm("d1_0"), drop_scale=dsc, sa_qkv=c["sa_qkv"], p_sa=c["p_sa"],
                    sa_o=c["sa_o"], t1_pre=c["t1_pre"], t1=c["t1"], m1=c["m1"], r1=c["r1"], caq=c["caq"],
                    N=w.N, S=S, K=self.K, H=H, add_xres=0 if self.bn else 1, B=B, Q=Q, heads=heads,
                    planes={k: pls.ptr(k) for k in pls.keys if not k.endswith("_t") and "dec_" not in k} if pls is not None else None)
            w.tables[key].launch()
            self._erank_fork(w)
            return
        key = ("fwd_pairs", bool(drop))
        if key not in w.tables:
            wi, bi = a.p(pl + "multihead_attn.in_proj_weight"), a.p(pl + "multihead_attn.in_proj_bias")
            t = 1 if H < 256 else 2
            g1 = ops.GemmGroup(GEMM_NT, [
                dict(a=w.h1, b=wv, c=w.vsw, c_row_xor=1),                              # V of the OTHER modality token
                dict(a=w.tgt0, b=a.p(pl + "self_attn.in_proj_weight"), c=c["sa_qkv"], a_add=qpos, a_add_mod=Q,
                     bias=a.p(pl + "self_attn.in_proj_bias"))], tile=t)
            g2 = ops.GemmGroup(GEMM_NT, [
                dict(a=w.h2, b=a.p(pre + "mlp.mlp.0.weight"), c=w.f1, bias=a.p(pre + "mlp.mlp.0.bias"), act=2, pre_out=w.u),
                dict(a=c["sa_o"], b=a.p(pl + "self_attn.out_proj.weight"), c=c["t1_pre"],
                     bias=a.p(pl + "self_attn.out_proj.bias"), drop_mask=self._dm2(dm("d1_0"), BQ, H), drop_scale=dsc)],
                tile=t)                                                             # layer 0: tgt = 0, no residual (:209)
            g3 = ops.GemmGroup(GEMM_NT, [
                dict(a=w.fused, b=a.p("fc_seg.weight"), c=w.seg, bias=a.p("fc_seg.bias")),          # (:228-232)
                dict(a=w.fused, b=wi[H:], c=c["cakv"], bias=bi[H:], a_add=pos, a_add_mod=S),      # k = v = memory + pos
                dict(a=c["t1"], b=wi[:H], c=c["caq"], a_add=qpos, a_add_mod=Q, bias=bi[:H])], tile=t)
            w.tables[key] = (g1, g2, g3)
        g1, g2, g3 = w.tables[key]
        g1.launch()
        gln = self._gln(2 * w.N, H) and self._gln(2 * w.N, 4 * H) and self._gln(BQ, 0)
        sa = dict(q=c["sa_qkv"][:, :H], k=c["sa_qkv"][:, H:2 * H], v=c["sa_qkv"][:, 2 * H:], probs=c["p_sa"], o=c["sa_o"], B=B,
                  heads=heads, Lq=Q, Lk=Q, dh=dh, drop_mask=dm("sa_p0"), drop_scale=dsc)
        ride = gln and self.ride_attention and ops.gemm_ln_mha_supported(heads, Q, Q, dh)
        if gln:      # attn.proj + x -> norm2 (transformerblock.py:131-132): the LayerNorm is the product's epilogue; the
            #          query self-attention core (parameters only) rides in the same launch
            ops.gemm_ln_fwd([dict(a=w.vsw, w=a.p(pre + "attn.proj.weight"), bias=a.p(pre + "attn.proj.bias"), res1=w.x0,
                                  pre=w.x1, gamma=a.p(pre + "norm2.weight"), beta=a.p(pre + "norm2.bias"), y=w.h2,
                                  mean=w.m2, rstd=w.r2)], mha=sa if ride else None)
        else:
            ops.gemm(GEMM_NT, w.vsw, a.p(pre + "attn.proj.weight"), w.x1, bias=a.p(pre + "attn.proj.bias"), res1=w.x0,
                     ws=self.ws)
        if not ride:
            ops.mha_core_fwd(sa["q"], sa["k"], sa["v"], sa["probs"], sa["o"], B, heads, Q, Q, dh, drop_mask=sa["drop_mask"],
                             drop_scale=dsc)
        if not gln:
            ops.layernorm_fwd(w.x1, a.p(pre + "norm2.weight"), a.p(pre + "norm2.bias"), w.h2, w.m2, w.r2)
        g2.launch()
        if gln:      # mlp fc2 + x (+ x_res) -> fuser.norm -> token mean (:92-94), and the decoder's norm1 as a plain job
            ops.gemm_ln_fwd([
                dict(a=w.f1, w=a.p(pre + "mlp.mlp.2.weight"), bias=a.p(pre + "mlp.mlp.2.bias"), res1=w.x1,
                     res2=None if self.bn else w.x0, pre=w.x3, gamma=a.p("fuser.norm.weight"),
                     beta=a.p("fuser.norm.bias"), y=w.y, mean=w.mf, rstd=w.rf, pair_out=w.fused),
                dict(a=None, pre=c["t1_pre"], gamma=a.p(pl + "norm1.weight"), beta=a.p(pl + "norm1.bias"), y=c["t1"],
                     mean=c["m1"], rstd=c["r1"])])
        else:
            ops.gemm(GEMM_NT, w.f1, a.p(pre + "mlp.mlp.2.weight"), w.x3, bias=a.p(pre + "mlp.mlp.2.bias"), res1=w.x1,
                     res2=None if self.bn else w.x0, ws=self.ws)          # (no x_res in the BN-blend variant)
            ops.layernorm_fwd_multi([                  # decoder norm1 and the fuser's final norm + token mean: one launch
                dict(x=c["t1_pre"], gamma=a.p(pl + "norm1.weight"), beta=a.p(pl + "norm1.bias"), y=c["t1"], mean=c["m1"],
                     rstd=c["r1"]),
                dict(x=w.x3, gamma=a.p("fuser.norm.weight"), beta=a.p("fuser.norm.bias"), y=w.y, mean=w.mf, rstd=w.rf,
                     pair_out=w.fused)])
        self._erank_fork(w)
        g3.launch()

    def _decoder_unfused(self, w, key_labels, dm, dsc, multi, main, s2, sa_block, paired=False):
        """The decoder composed from the GEMM / attention / LayerNorm entry points (any shape)."""
        a, H, Q, K, heads, dh = self.arena, self.H, self.Q, self.K, self.heads, self.dh
        B, S, N, BQ = w.B, w.S, w.N, w.BQ
        pos = a.p("pos_embedding")[0, :S]
        tgt = None
        for l in range(self.L):
            c, pl = w.layers[l], f"transformer.decoder.layers.{l}."
            wi, bi = a.p(pl + "multihead_attn.in_proj_weight"), a.p(pl + "multihead_attn.in_proj_bias")
            # key = value = memory + pos (:300-302): the broadcast add is the GEMM's A-operand prologue
            if l > 0:                      # (layer 0: in the segmentation head's launch, see forward_finish)
                ops.gemm(GEMM_NT, w.fused, wi[H:], c["cakv"], a_add=pos, a_add_mod=S, bias=bi[H:], ws=self.ws)
            if l == 0:
                if multi:
                    main.wait_stream(s2)
            else:
                sa_block(l, tgt, self.ws)
            # (paired: the layer-0 sub-layer ran inside _forward_paired)
            w._dec_deferred = False
            if paired and self._dec_chain_ok(w):
                # cross-attention core -> out_proj + norm2 -> FFN -> t3_pre: one launch (csrc/decoder_chain.hip) -- or none
                # here at all: in a training step (defer_tail) the launch inside losses() runs this forward half, the tail,
                # the losses and the backward half together
                drop_now = self.last_drop_flag
                w._tail_done = False
                w._tail_deferred = bool(self.use_fused_tail and self.defer_tail and self._fw["mode"] == "train" and
                                        hasattr(w, "glayers") and ops.tail_losses_supported(H, self.K + 1, Q, BQ))
                w._dec_key_labels = key_labels
                if w._tail_deferred:
                    w._dec_deferred = True
                    return
                self._dec_chain(w, drop_now).launch(1, key_label=key_labels)
                if self.use_fused_tail:
                    ops.decoder_tail_fwd(c["t3_pre"], a.p(pl + "norm3.weight"), a.p(pl + "norm3.bias"),
                                         a.p("transformer.decoder.norm.weight"), a.p("transformer.decoder.norm.bias"),
                                         self.w_head, self.b_head, c["t3"], c["m3"], c["r3"], w.tgtF, w.mF, w.rF, w.actdur)
                    return
                ops.layernorm_fwd(c["t3_pre"], a.p(pl + "norm3.weight"), a.p(pl + "norm3.bias"), c["t3"], c["m3"], c["r3"])
                tgt = c["t3"]
                continue
            ops.mha_core_fwd(c["caq"], c["cakv"][:, :H], c["cakv"][:, H:], c["p_ca"], c["ca_o"], B, heads, Q, S, dh,
                             key_labels=key_labels, pad_idx=self.pad_idx, drop_mask=dm(f"ca_p{l}"), drop_scale=dsc)
            if self._gln(BQ, H):      # out_proj -> dropout -> + tgt -> norm2 (transformer.py:304-306): one launch
                ops.gemm_ln_fwd([dict(a=c["ca_o"], w=a.p(pl + "multihead_attn.out_proj.weight"),
                                      bias=a.p(pl + "multihead_attn.out_proj.bias"),
                                      drop_mask=self._dm2(dm(f"d2_{l}"), BQ, H), drop_scale=dsc, res1=c["t1"],
                                      pre=c["t2_pre"], gamma=a.p(pl + "norm2.weight"), beta=a.p(pl + "norm2.bias"),
                                      y=c["t2"], mean=c["m2"], rstd=c["r2"])])
            else:
                ops.gemm(GEMM_NT, c["ca_o"], a.p(pl + "multihead_attn.out_proj.weight"), c["t2_pre"],
                         bias=a.p(pl + "multihead_attn.out_proj.bias"), drop_mask=self._dm2(dm(f"d2_{l}"), BQ, H),
                         drop_scale=dsc, res1=c["t1"], ws=self.ws)
                ops.layernorm_fwd(c["t2_pre"], a.p(pl + "norm2.weight"), a.p(pl + "norm2.bias"), c["t2"], c["m2"], c["r2"])
            ops.gemm(GEMM_NT, c["t2"], a.p(pl + "linear1.weight"), c["ff1"], bias=a.p(pl + "linear1.bias"), act=1,
                     drop_mask=self._dm2(dm(f"ff_{l}"), BQ, 4 * H), drop_scale=dsc, ws=self.ws)
            ops.gemm(GEMM_NT, c["ff1"], a.p(pl + "linear2.weight"), c["t3_pre"], bias=a.p(pl + "linear2.bias"),
                     drop_mask=self._dm2(dm(f"d3_{l}"), BQ, H), drop_scale=dsc, res1=c["t2"], ws=self.ws)
            if l == self.L - 1 and self.use_fused_tail:
                # last norm3 + decoder.norm (:182-183) + anticipation heads (:219-226, fc | fc_len = one [K+1, H] product)
                w._tail_done = False
                w._tail_deferred = bool(self.defer_tail and self._fw["mode"] == "train" and hasattr(w, "glayers")
                                        and ops.tail_losses_supported(H, self.K + 1, Q, BQ))
                if w._tail_deferred:
                    return                          # runs inside losses()
                ops.decoder_tail_fwd(c["t3_pre"], a.p(pl + "norm3.weight"), a.p(pl + "norm3.bias"),
                                     a.p("transformer.decoder.norm.weight"), a.p("transformer.decoder.norm.bias"),
                                     self.w_head, self.b_head, c["t3"], c["m3"], c["r3"], w.tgtF, w.mF, w.rF, w.actdur)
                return
            ops.layernorm_fwd(c["t3_pre"], a.p(pl + "norm3.weight"), a.p(pl + "norm3.bias"), c["t3"], c["m3"], c["r3"])
            tgt = c["t3"]
        ops.layernorm_fwd(tgt, a.p("transformer.decoder.norm.weight"), a.p("transformer.decoder.norm.bias"), w.tgtF,
                          w.mF, w.rF)
        # ---- anticipation heads (:219-226): fc and fc_len as one [K+1, H] GEMM
        ops.gemm(GEMM_NT, w.tgtF, self.w_head, w.actdur, bias=self.b_head, ws=self.ws)

    def _decoder_fused(self, w, key_labels, drop, dsc):
        """One launch per decoder layer (decoder.hip): a workgroup keeps one clip's layer in LDS."""
        a, H, Q, K, heads = self.arena, self.H, self.Q, self.K, self.heads
        B, S = w.B, w.S
        key = ("dec_fwd", bool(drop))
        if key not in w.tables:
            tabs = []
            for l in range(self.L):
                c, pl = w.layers[l], f"transformer.decoder.layers.{l}."
                p = lambda n: a.p(pl + n)         # noqa: E731
                dk = (lambda k: w.drop[k]) if drop else (lambda k: None)
                last = l == self.L - 1
                t = [p("self_attn.in_proj_weight"), p("self_attn.in_proj_bias"), p("self_attn.out_proj.weight"),
                     p("self_attn.out_proj.bias"), p("norm1.weight"), p("norm1.bias"), p("multihead_attn.in_proj_weight"),
                     p("multihead_attn.in_proj_bias"), p("multihead_attn.out_proj.weight"), p("multihead_attn.out_proj.bias"),
                     p("norm2.weight"), p("norm2.bias"), p("linear1.weight"), p("linear1.bias"), p("linear2.weight"),
                     p("linear2.bias"), p("norm3.weight"), p("norm3.bias"),
                     w.fused, a.p("pos_embedding")[0, :S], a.p("query_embed.weight"),
                     None if l == 0 else w.layers[l - 1]["t3"], None,
                     dk(f"sa_p{l}"), dk(f"ca_p{l}"), dk(f"d1_{l}"), dk(f"d2_{l}"), dk(f"d3_{l}"), dk(f"ff_{l}"),
                     c["sa_qkv"], c["p_sa"], c["sa_o"], c["t1_pre"], c["t1"], c["m1"], c["r1"], c["caq"], c["cakv"], c["p_ca"],
                     c["ca_o"], c["t2_pre"], c["t2"], c["m2"], c["r2"], c["ff1"], c["t3_pre"], c["t3"], c["m3"], c["r3"]]
                if last:
                    t += [a.p("transformer.decoder.norm.weight"), a.p("transformer.decoder.norm.bias"), w.tgtF, w.mF, w.rF,
                          self.w_head, self.b_head, w.actdur]
                else:
                    t += [None] * 8
                tabs.append(ops.PtrTable(t))
            w.tables[key] = tabs
        for l, tab in enumerate(w.tables[key]):
            tab.arr[22] = key_labels.data_ptr() if key_labels is not None else None
            ops.decoder_layer_fwd(tab, B, S, Q, H, heads, self.pad_idx, dsc, K + 1)

    @staticmethod
    def _dm2(m, rows, cols):
        return None if m is None else m.view(rows, cols)

    # ------------------------------------------------------------------------------------------------------
    def losses(self, past_label, target, target_dur, with_grad=True, val_mode=False, tick=False):
        """The 3 losses + counters of train_proposed_depth.py:171-213 in one launch; fills d_seg / d_actdur.
        tick=True: the launch also advances the optimiser's step counter (and the dropout offset when dropout ran), so
        the following adamw(..., ticked=True) needs no launch of its own for that."""
        w = self.last["w"]
        K = self.K
        # a reduction job left pending by an earlier deferred losses() belongs to THAT step: it may only be consumed by the
        # adamw() that follows it directly -- never by the optimiser step of another forward (a backward that raised, a flow
        # that switched to the undeferred reduction)
        self._loss_pending = None
        ta = self.step_t if tick else None
        tb = self.drop_offset if (tick and self.last["drop"]) else None
        if getattr(w, "_tail_deferred", False):
            w._tail_deferred = False
            a, Lm = self.arena, self.L - 1
            c, pl = w.layers[Lm], f"transformer.decoder.layers.{Lm}."
            tail = dict(x=c["t3_pre"], g3=a.p(pl + "norm3.weight"), b3=a.p(pl + "norm3.bias"),
                        gF=a.p("transformer.decoder.norm.weight"), bF=a.p("transformer.decoder.norm.bias"),
                        w_head=self.w_head, b_head=self.b_head, t3=c["t3"], m3=c["m3"], r3=c["r3"], tgtF=w.tgtF, mF=w.mF,
                        rF=w.rF, out=w.actdur)
            dec = None
            if getattr(w, "_dec_deferred", False):
                w._dec_deferred = False
                dec = self._dec_chain(w, self.last["drop"])
                if not (with_grad and not val_mode):         # (no gradient wanted: only the forward half is due)
                    dec.launch(1, key_label=w._dec_key_labels)
                    dec = None
            if with_grad and not val_mode:
                # forward tail + losses + backward tail: one launch (losses.hip: tail_losses_kernel) -- with `dec` the
                # decoder's whole query side around it as well (decoder_chain.hip, phases 7)
                drop = self.last["drop"]
                defer = bool(self.defer_loss_reduce and tick)
                ops.decoder_tail_losses(**tail, seg=w.seg, past_label=past_label, target=target, target_dur=target_dur,
                                        B=w.B, S=w.S, Q=self.Q, K=K, pad_idx=self.pad_idx, exclude_idx=EXCLUDE_CLASS_IDX,
                                        dur_den=self.dur_den, grad_scale=1.0, d_seg=w.d_seg, d_out=w.d_actdur, loss_out=w.loss,
                                        counts=w.counts, tick_a=ta, tick_b=tb,
                                        drop=w.drop[f"d3_{Lm}"] if drop else None, drop_scale=1.0 / (1.0 - DROP_P),
                                        dx=w.glayers[Lm]["t3pre"], dx2=w.glayers[Lm]["ff2"], wsF=w.lnp["final"],
                                        ws3=w.lnp[f"d3_{Lm}"], ws=w.loss_ws, defer_finalize=defer, chain=dec,
                                        chain_key_label=w._dec_key_labels if dec is not None else None)
                w._tail_done = True
                w._dec_bwd_done = dec is not None
                acc = self.loss_acc if self.loss_acc is not None else (None, None)
                self._loss_pending = ops.loss_finalize_job(w.loss_ws, w.B, w.S, self.Q, True, self.dur_den, w.loss, w.counts,
                                                           acc_loss=acc[0], acc_counts=acc[1]) if defer else None
                return w.loss, w.counts
            ops.decoder_tail_fwd(tail["x"], tail["g3"], tail["b3"], tail["gF"], tail["bF"], tail["w_head"], tail["b_head"],
                                 tail["t3"], tail["m3"], tail["r3"], tail["tgtF"], tail["mF"], tail["rF"], tail["out"])
        ops.losses_fwd_bwd(None if val_mode else w.seg, w.actdur[:, :K], w.actdur[:, K:], K + 1, past_label, target,
                           target_dur, w.B, w.S, self.Q, K, self.pad_idx, EXCLUDE_CLASS_IDX, w.loss, w.counts,
                           val_mode=val_mode, dur_den=self.dur_den,
                           d_seg=w.d_seg if with_grad else None, d_act=w.d_actdur[:, :K] if with_grad else None,
                           d_dur=w.d_actdur[:, K:] if with_grad else None, ld_ddur=K + 1, ws=w.loss_ws, tick_a=ta,
                           tick_b=tb)
        return w.loss, w.counts

    # ------------------------------------------------------------------------------------------------------
    def backward(self, d_seg=None, d_actdur=None, fused_adamw=None, adamw_next=False):
        """Adjoint of forward(); gradients land in the grad arena (written, not accumulated).
        adamw_next: the caller enqueues adamw() right after (train_step): the parameter-gradient branch (overlap_param_tail)
        is then left open for the AdamW of its parameters to continue on; otherwise it is joined before returning.
        fused_adamw: dict(lr, weight_decay[, betas, eps, grad_scale]) -> depth_projection.weight (86 % of the model) is
        updated INSIDE its weight-gradient GEMM (the gradient is never written); follow with adamw(..., skip_depth=True).
        Only valid when that gradient needs no exchange (one GPU, or the pixel-sharded projection)."""
        self.prepare_fused_adamw(fused_adamw)
        self._overlap_tail_now = bool((self.overlap_param_tail or (self.auto_side_stream and self.H >= 512)) and
                                      self.last["tp"] is None and self.tp is None and
                                      self.grad_hook is None and (self._adam is None or self.H >= 512) and
                                      "depth_projection.weight" in self.arena.offsets)
        try:
            self.backward_main(d_seg, d_actdur)
        finally:
            self._overlap_tail_now = False
        if self.last["tp"] is not None:         # before the small bucket: the weight gradient waits for this one
            self.last["tp"].exchange_backward(self.last["w"])
        if self.grad_hook is not None:
            self.grad_hook("small_ready")
        self.backward_depth_wgrad()
        if not adamw_next:
            self._join_tail()
        if self.grad_hook is not None:
            self.grad_hook("big_ready")

    def depth_adamw_fusable(self):
        """True when depth_projection.weight can be updated inside its weight-gradient kernel at a gain: one rank (its gradient
        needs no exchange) and the product runs on the tiled bf16x3 TN kernel (tiles 10 / 12: more than 128 token rows or hidden units),
        whose epilogue then streams parameter and moments while the rest of the chip multiplies -- the flat AdamW launch no
        longer reads and re-writes 86 % of the model (cfg4's per-GPU shape 1.127 -> 1.082 ms, cfg5's 1.644 -> 1.553; at the
        headline shape the product runs on the panel kernel, where fusing was measured neutral, so it stays off there)."""
        st = self.last
        if (not self.fuse_depth_adamw or st is None or self.tp is not None or st["tp"] is not None or
                self.grad_hook is not None or self.depth_prec != 1):
            return False
        w = st["w"]
        if not hasattr(w, "_depth_tile"):
            w._depth_tile = ops.gemm_planned_tile(GEMM_TN, w.d_dep_pre, st["x_dep"], self.arena.p("depth_projection.weight"),
                                                  prec=self.depth_prec)
        return w._depth_tile in (10, 12)

    def prepare_fused_adamw(self, cfg):
        """cfg: None or dict(lr, weight_decay[, betas, eps, grad_scale]); consumed by backward_depth_wgrad()."""
        self._adam = None
        if cfg is not None:
            self.set_lr(cfg["lr"])
            b = cfg.get("betas", (0.9, 0.999))
            self._adam = dict(lr_t=self.lr_t, step_t=self.step_t, beta1=b[0], beta2=b[1], eps=cfg.get("eps", 1e-8),
                              weight_decay=cfg["weight_decay"], grad_scale=cfg.get("grad_scale", 1.0))

    def backward_depth_wgrad(self):
        """depth_projection.weight gradient [H, 50176] = d_dep_pre^T . depth -- the last and largest kernel of the
        backward (81 % of the gradient bytes at H=128); everything else is complete before it starts."""
        st = self.last
        adam = getattr(self, "_adam", None)
        if st["tp"] is not None:                # this rank's pixel columns, summed over every rank's clips
            st["tp"].wgrad(st["w"], self.ws, adam)
            return
        if adam is not None:
            a = self.arena
            o, n, shp = a.offsets["depth_projection.weight"]
            ops.gemm(GEMM_TN, st["w"].d_dep_pre, st["x_dep"], a.p("depth_projection.weight"), ws=self.ws, prec=self.depth_prec,
                     adam=dict(adam, m=a.exp_avg[o:o + n].view(shp), v=a.exp_avg_sq[o:o + n].view(shp)))
            return
        ops.gemm(GEMM_TN, st["w"].d_dep_pre, st["x_dep"], self.arena.g("depth_projection.weight"), ws=self.ws,
                 prec=self.depth_prec)

    def _build_groups(self, w):
        """Once per shape: every weight gradient whose operands live in the persistent workspace becomes one problem of
        a grouped TN GEMM (bias gradient fused), every LayerNorm parameter reduction one job of a batched finalize."""
        a, H, Q, S, N, BQ = self.arena, self.H, self.Q, w.S, w.N, w.BQ
        qpos = a.p("query_embed.weight")
        pos = a.p("pos_embedding")[0, :S]
        pre = "fuser.blocks.0."
        P = []

        # (hidden >= 512: taking the 0.5 - 2 GFLOP weight gradients out of this launch was measured twice and reverted --
        #  through the planner's fp32 tiles cfg4's per-GPU shape went 1.179 -> 1.212 ms, cfg5's 1.690 -> 1.721; through the
        #  bf16x3 TN kernel (128 x 128 tiles: 64 workgroups for a [2048, 512] gradient, a quarter of the chip) 1.241 / 1.685)
        def add(dy, x, gw, gb=None, b_add=None, b_mod=0):
            P.append(dict(a=dy, b=x, c=gw, bias_grad=gb, b_add=b_add, b_add_mod=b_mod))
        add(w.d_actdur, w.tgtF, self.gw_head, self.gb_head)
        add(w.d_seg, w.fused, a.g("fc_seg.weight"), a.g("fc_seg.bias"))
        for l in range(self.L):
            c, gl, pl = w.layers[l], w.glayers[l], f"transformer.decoder.layers.{l}."
            g = lambda n: a.g(pl + n)         # noqa: E731
            tgt_in = w.tgt0 if l == 0 else w.layers[l - 1]["t3"]
            gwi, gbi = g("multihead_attn.in_proj_weight"), g("multihead_attn.in_proj_bias")
            add(gl["ff2"], c["ff1"], g("linear2.weight"), g("linear2.bias"))
            add(gl["ff1"], c["t2"], g("linear1.weight"), g("linear1.bias"))
            add(gl["cap"], c["ca_o"], g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias"))
            add(gl["cakv"], w.fused, gwi[H:], gbi[H:], pos, S)
            add(gl["caq"], c["t1"], gwi[:H], gbi[:H], qpos, Q)
            add(gl["sap"], c["sa_o"], g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"))
            add(gl["saqkv"], tgt_in, g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"), qpos, Q)
        add(w.d_x3, w.f1, a.g(pre + "mlp.mlp.2.weight"), a.g(pre + "mlp.mlp.2.bias"))
        add(w.d_u, w.h2, a.g(pre + "mlp.mlp.0.weight"), a.g(pre + "mlp.mlp.0.bias"))
        add(w.d_x1, w.vsw, a.g(pre + "attn.proj.weight"), a.g(pre + "attn.proj.bias"))
        add(w.d_v, w.h1, a.g(pre + "attn.qkv.weight")[2 * H:], None)          # rows [0,2H) (Q,K) stay exactly zero
        # input_embed's weight gradient reads the step's RGB batch: operand B is re-pointed per step (GemmGroup.set_b)
        w.rgb_wgrad_idx = len(P)
        add(w.d_rgb_pre, self.last["x_rgb"], a.g("input_embed.weight"), a.g("input_embed.bias"))
        w.wgrad_group = ops.GemmGroup(GEMM_TN, P, tile=1 if H < 256 else 2)
        w.wgrad_problems, w.wgrad_plus = P, {}
        J = [(w.lnp["final"], BQ, H, a.g("transformer.decoder.norm.weight"), a.g("transformer.decoder.norm.bias")),
             (w.lnp["nf"], 2 * N, H, a.g("fuser.norm.weight"), a.g("fuser.norm.bias")),
             (w.lnp["n2"], 2 * N, H, a.g(pre + "norm2.weight"), a.g(pre + "norm2.bias")),
             (w.lnp["n1"], 2 * N, H, a.g(pre + "norm1.weight"), a.g(pre + "norm1.bias")),
             (w.lnp["dep"], N, H, a.g("depth_layernorm.weight"), a.g("depth_layernorm.bias"))]
        for l in range(self.L):
            pl = f"transformer.decoder.layers.{l}."
            for k in (1, 2, 3):
                J.append((w.lnp[f"d{k}_{l}"], BQ, H, a.g(pl + f"norm{k}.weight"), a.g(pl + f"norm{k}.bias")))
        w.ln_group = ops.LnFinalizeGroup(J)
        Js = [(w.lnp_seam["n1"], -N, H) + j[3:] if j[0] is w.lnp["n1"] else
              ((w.lnp_seam["dep"], -N, H) + j[3:] if j[0] is w.lnp["dep"] else j) for j in J]
        w.ln_group_seam = ops.LnFinalizeGroup(Js)
        Jc = None
        if self._chain_shape_ok(w):
            # the chain backward (csrc/fuser_chain.hip) leaves one (dgamma, dbeta) pair per 4 rows for the fuser's norm /
            # norm2 and the decoder's norm1, and one per frame (the seam's layout) for norm1 / the depth LayerNorm
            f = lambda n_: torch.empty(n_, dtype=torch.float32, device=self.device)     # noqa: E731
            w.chain_parts = dict(nf=f(2 * N // 4 * 2 * H), n2=f(2 * N // 4 * 2 * H), d1=f(BQ // 4 * 2 * H))
            w.d_extra = torch.empty(N, H, dtype=torch.float32, device=self.device)
            rep = [(w.lnp["nf"], w.chain_parts["nf"], -(2 * N // 4)), (w.lnp["n2"], w.chain_parts["n2"], -(2 * N // 4)),
                   (w.lnp["n1"], w.lnp_seam["n1"], -N), (w.lnp["dep"], w.lnp_seam["dep"], -N),
                   (w.lnp["d1_0"], w.chain_parts["d1"], -(BQ // 4))]

            def swap(j):
                for old_, new_, rows_ in rep:
                    if j[0] is old_:
                        return (new_, rows_, H) + j[3:]
                return j
            Jc = [swap(j) for j in J]
        Jb = None
        if self.bn:                            # norm1's partials come from r3d_bn_blend_bwd (one per frame)
            Jb = [(w.lnp_seam["n1"], -N, H) + j[3:] if j[0] is w.lnp["n1"] else j for j in J]
            t = w.bn_terms
            w.bn_sums = ops.RowsumGroup([(t[0], None, 1, a.g("fuser.bn_rgb.bias").view(1, H)),
                                         (t[1], None, 1, a.g("fuser.bn_rgb.weight").view(1, H)),
                                         (t[2], None, 1, a.g("fuser.bn_depth.bias").view(1, H)),
                                         (t[3], None, 1, a.g("fuser.bn_depth.weight").view(1, H)),
                                         (t[4], None, 1, a.g("fuser.alpha").view(1, H))])
        R = [(w.d_fused, None, S, a.g("pos_embedding")[0, :S]),
             (w.d_dep_pre, None, 1, a.g("depth_projection.bias").view(1, H)),
             (w.glayers[self.L - 1]["caqin"], w.glayers[self.L - 1]["sain"], Q, a.g("query_embed.weight"))]
        w.rowsum_group = ops.RowsumGroup(R)
        # The LayerNorm partial sums are column sums as well: when a site's (weight, bias) gradients are adjacent in the
        # arena (they are: same size class, declaration order) its finalize is one more job of the row-sum launch.

        def as_rowsum(job):
            part, rows, Hh, dg, db = job
            if db.data_ptr() != dg.data_ptr() + 4 * Hh:
                return None
            if rows < 0:
                blocks = -rows
            else:
                rpb = max(4, ((rows + 255) // 256 + 3) // 4 * 4)
                blocks = (rows + rpb - 1) // rpb
                if blocks <= 1:
                    return None                 # layernorm_bwd wrote the final values itself
            o = (dg.data_ptr() - a.grads.data_ptr()) // 4
            return (part[:blocks * 2 * Hh].view(blocks, 2 * Hh), None, 1, a.grads[o:o + 2 * Hh].view(1, 2 * Hh))
        w.tail_groups, w.tail_jobs = {}, {}
        for name, jobs in ((("plain", J), ("seam", Js)) + ((("bn", Jb),) if Jb is not None else ()) +
                           ((("chain", Jc),) if Jc is not None else ())):
            conv = [as_rowsum(j) for j in jobs]
            if all(c is not None for c in conv):
                w.tail_groups[name] = ops.RowsumGroup(R + conv)
                w.tail_jobs[name] = R + conv

    def _wgrad_with_sums(self, w, name):
        """The grouped weight-gradient launch with the row-sum jobs of tail group `name` folded in as TN problems:
        dst[r, :] = sum over rows with row % mod == r of (src1 + src2) = sel^T . (src1 + src2), sel[row, row % mod] = 1
        (exact products, fp32 accumulation).  None when the table would not fit the kernel's 32-problem argument."""
        if name in w.wgrad_plus:
            return w.wgrad_plus[name]
        g = None
        jobs = w.tail_jobs.get(name)
        if jobs is not None and len(w.wgrad_problems) + len(jobs) <= 32 and self.H < 256:
            extra = []
            for s1, s2, mod, dst in jobs:
                rows = s1.shape[0]
                key = (rows, mod)
                if key not in self._sel:
                    sel = torch.zeros(rows, mod, dtype=torch.float32, device=self.device)
                    sel[torch.arange(rows, device=self.device), torch.arange(rows, device=self.device) % mod] = 1.0
                    self._sel[key] = sel
                extra.append(dict(a=self._sel[key], b=s1, c=dst, b_add=s2, b_add_mod=rows if s2 is not None else 0))
            g = ops.GemmGroup(GEMM_TN, w.wgrad_problems + extra, tile=1)
        w.wgrad_plus[name] = g
        return g

    def backward_main(self, d_seg=None, d_actdur=None):
        """Everything of the backward except depth_projection.weight.
        The chain of input gradients (the critical path) is a sequence of dependent, latency-bound launches; nothing
        that only feeds a parameter gradient stays on it: all weight/bias gradients run as ONE grouped GEMM launch
        and all LayerNorm parameter reductions as ONE batched launch after the chain.  With use_side_stream the
        self-attention sub-layer of the decoder queries (which feeds only parameter gradients for a one-layer
        decoder) runs on a second HIP stream -- a parallel branch under hipGraph capture."""
        st = self.last
        w, a, H, Q, K, heads, dh, ws = st["w"], self.arena, self.H, self.Q, self.K, self.heads, self.dh, self.ws
        B, S, N, BQ = w.B, w.S, w.N, w.BQ
        ext_grads = d_seg is not None or d_actdur is not None       # (autograd path: gradients handed in by the caller)
        if d_seg is not None and d_seg.data_ptr() != w.d_seg.data_ptr():
            w.d_seg.copy_(d_seg)
        if d_actdur is not None and d_actdur.data_ptr() != w.d_actdur.data_ptr():
            w.d_actdur.copy_(d_actdur)
        d_seg, d_actdur = w.d_seg, w.d_actdur
        if not hasattr(w, "wgrad_group"):
            self._build_groups(w)
        drop = st["drop"]
        dsc = 1.0 / (1.0 - DROP_P)
        dm = (lambda k, r, c: w.drop[k].view(r, c)) if drop else (lambda k, r, c: None)
        dmf = (lambda k: w.drop[k]) if drop else (lambda k: None)
        main = torch.cuda.current_stream()
        multi = self._multi_stream()
        s2 = self.side2 if multi else main
        ws2 = self.ws_side2 if multi else ws
        joined = True

        def ln_bwd(site, dy, x, mean, rstd, gname, bname, dx, **kw):
            ops.layernorm_bwd(dy, x, mean, rstd, a.p(gname), a.p(bname), dx, a.g(gname), a.g(bname), partial=w.lnp[site],
                              **kw)

        last = w.layers[-1]
        tail = self.use_fused_tail
        paired = bool(st.get("paired"))
        if tail:
            # ---- heads' input gradient + decoder.norm backward + the last norm3 backward: one launch.  The segmentation
            # head's input gradient (only needed at the fuser's norm) rides in a later group (paired) or goes alone.
            Lm, plm = self.L - 1, f"transformer.decoder.layers.{self.L - 1}."
            if getattr(w, "_tail_done", False) and not ext_grads:
                w._tail_done = False                 # losses() already ran the tail's backward (r3d_decoder_tail_losses)
            else:
                ops.decoder_tail_bwd(w.d_actdur, self.w_head, last["t3"], w.mF, w.rF,
                                     a.p("transformer.decoder.norm.weight"), last["t3_pre"], last["m3"], last["r3"],
                                     a.p(plm + "norm3.weight"), None if not drop else w.drop[f"d3_{Lm}"], dsc,
                                     w.glayers[Lm]["t3pre"], w.glayers[Lm]["ff2"], a.g("transformer.decoder.norm.weight"),
                                     a.g("transformer.decoder.norm.bias"), a.g(plm + "norm3.weight"),
                                     a.g(plm + "norm3.bias"), w.lnp["final"], w.lnp[f"d3_{Lm}"])
            if not paired:
                ops.gemm(GEMM_NN, d_seg, a.p("fc_seg.weight"), w.d_fused2, ws=ws)
        else:
            # ---- heads: both input gradients in one launch (the segmentation one is only needed at the fuser's norm)
            if not hasattr(w, "head_group"):
                w.head_group = ops.GemmGroup(GEMM_NN, [dict(a=w.d_actdur, b=self.w_head, c=w.d_tgtF),
                                                       dict(a=w.d_seg, b=a.p("fc_seg.weight"), c=w.d_fused2)],
                                             tile=1 if H < 256 else 2)
            w.head_group.launch()
            # ---- decoder
            ln_bwd("final", w.d_tgtF, last["t3"], w.mF, w.rF, "transformer.decoder.norm.weight",
                   "transformer.decoder.norm.bias", w.d_t)
        dy, dy2 = w.d_t, None                 # gradient w.r.t. t3 of the current layer (= dy + dy2)
        first_fused = True
        dec_done = bool(getattr(w, "_dec_bwd_done", False)) and not ext_grads     # losses() ran the decoder's backward half
        w._dec_bwd_done = False
        dec_chain = bool(paired and self._dec_chain_ok(w))
        for l in reversed(range(self.L)):
            c, gl, pl = w.layers[l], w.glayers[l], f"transformer.decoder.layers.{l}."
            p = lambda n: a.p(pl + n)         # noqa: E731
            if dec_done:
                break
            # norm3 -> (t2 residual, FFN)
            if not (tail and l == self.L - 1):
                ln_bwd(f"d3_{l}", dy, c["t3_pre"], c["m3"], c["r3"], pl + "norm3.weight", pl + "norm3.bias", gl["t3pre"],
                       dy2=dy2, dx2=gl["ff2"], drop_mask=dm(f"d3_{l}", BQ, H), drop_scale=dsc)
            if dec_chain:                      # FFN / norm2 / out_proj input gradients + attention core backward: one launch
                self._dec_chain(w, drop).launch(4)
                break
            ops.gemm(GEMM_NN, gl["ff2"], p("linear2.weight"), gl["ff1"], drop_mask=dm(f"ff_{l}", BQ, 4 * H),
                     drop_scale=dsc, aux=c["ff1"], mul=1, ws=ws)
            split = self.split_k4h and H < 256 and H % 4 == 0
            if split:                          # two half-K problems, one launch; norm2's backward adds them (dy + dy2)
                key = ("l1_dgrad", l)
                if key not in w.tables:
                    w1 = p("linear1.weight")
                    w.tables[key] = ops.GemmGroup(GEMM_NN, [
                        dict(a=gl["ff1"][:, :2 * H], b=w1[:2 * H], c=gl["t2"], res1=gl["t3pre"]),
                        dict(a=gl["ff1"][:, 2 * H:], b=w1[2 * H:], c=gl["t2b"])], tile=1)
                w.tables[key].launch()
            else:
                ops.gemm(GEMM_NN, gl["ff1"], p("linear1.weight"), gl["t2"], res1=gl["t3pre"], ws=ws)
            # norm2 -> (t1 residual, cross attention)
            ln_bwd(f"d2_{l}", gl["t2"], c["t2_pre"], c["m2"], c["r2"], pl + "norm2.weight", pl + "norm2.bias", gl["t2pre"],
                   dy2=gl["t2b"] if split else None, dx2=gl["cap"], drop_mask=dm(f"d2_{l}", BQ, H), drop_scale=dsc)
            ops.gemm(GEMM_NN, gl["cap"], p("multihead_attn.out_proj.weight"), gl["cao"], ws=ws)
            ops.mha_core_bwd(c["caq"], c["cakv"][:, :H], c["cakv"][:, H:], c["p_ca"], gl["cao"], gl["caq"],
                             gl["cakv"][:, :H], gl["cakv"][:, H:], B, heads, Q, S, dh, drop_mask=dmf(f"ca_p{l}"),
                             drop_scale=dsc)
            wi = p("multihead_attn.in_proj_weight")
            if st.get("paired"):
                break                          # one layer, one stream: continued below with paired launches
            # ---- query path of the cross attention + the self-attention sub-layer (branch s2)
            if multi:
                s2.wait_stream(main)
                joined = False
            with torch.cuda.stream(s2):
                ops.gemm(GEMM_NN, gl["caq"], wi[:H], gl["caqin"], ws=ws2)
                # norm1: d t1 = caqin (query path) + t2pre (residual into t2_pre)
                ln_bwd(f"d1_{l}", gl["caqin"], c["t1_pre"], c["m1"], c["r1"], pl + "norm1.weight", pl + "norm1.bias",
                       gl["t1pre"], dy2=gl["t2pre"], dx2=gl["sap"], drop_mask=dm(f"d1_{l}", BQ, H), drop_scale=dsc)
                ops.gemm(GEMM_NN, gl["sap"], p("self_attn.out_proj.weight"), gl["sao"], ws=ws2)
                ops.mha_core_bwd(c["sa_qkv"][:, :H], c["sa_qkv"][:, H:2 * H], c["sa_qkv"][:, 2 * H:], c["p_sa"], gl["sao"],
                                 gl["saqkv"][:, :H], gl["saqkv"][:, H:2 * H], gl["saqkv"][:, 2 * H:], B, heads, Q, Q, dh,
                                 drop_mask=dmf(f"sa_p{l}"), drop_scale=dsc)
                ops.gemm(GEMM_NN, gl["saqkv"], p("self_attn.in_proj_weight"), gl["sain"], ws=ws2)
            ops.gemm(GEMM_NN, gl["cakv"], wi[H:], w.d_fused, accumulate=not first_fused, ws=ws)   # d (memory + pos)
            first_fused = False
            if l > 0:                          # d t3 of layer l-1 = sain (through the queries) + t1pre (residual)
                if multi:
                    main.wait_stream(s2)
                    joined = True
                dy, dy2 = gl["sain"], gl["t1pre"]
        # ---- fuser; d(memory) = decoder part (d_fused, kept for the positional-embedding gradient :190) + seg head part
        pre = "fuser.blocks.0."
        chain = bool(st.get("paired") and self._chain_ok(w))
        if chain:
            # everything from the decoder's memory-side input gradients to the two embeddings' pre-activation gradients,
            # and the query-side branch, in ONE launch (csrc/fuser_chain.hip)
            c, gl, pl = w.layers[0], w.glayers[0], "transformer.decoder.layers.0."
            er = bool(st.get("erank"))
            bf3 = bool(self.chain_bf3 and K <= 32)
            key = ("bwd_chain", bool(drop), er, bf3)
            if key not in w.tables:
                pls = self.chain_planes() if bf3 else None
                wi0 = a.p(pl + "multihead_attn.in_proj_weight")
                w.tables[key] = ops.FuserChainBwd(
                    d_cakv=gl["cakv"], d_seg=w.d_seg, d_extra=w.d_extra if er else None, wkv=wi0[H:], wseg=a.p("fc_seg.weight"),
                    x3=w.x3, mf=w.mf, rf=w.rf, gf=a.p("fuser.norm.weight"), w2=a.p(pre + "mlp.mlp.2.weight"), u=w.u,
                    w1=a.p(pre + "mlp.mlp.0.weight"), x1=w.x1, m2=w.m2, r2=w.r2, g2=a.p(pre + "norm2.weight"),
                    wproj=a.p(pre + "attn.proj.weight"), wv=a.p(pre + "attn.qkv.weight")[2 * H:], x0=w.x0, m1=w.m1, r1=w.r1,
                    g1n=a.p(pre + "norm1.weight"), drop_x0=dmf("x0"), m_rgb=st["mask"][0], m_dep=st["mask"][1], rgb=w.rgb,
                    dep_pre=w.dep_pre, mean_d=w.mean_d, rstd_d=w.rstd_d, lnd_g=a.p("depth_layernorm.weight"),
                    lnd_b=a.p("depth_layernorm.bias"), d_fused=w.d_fused, d_x3=w.d_x3, d_u=w.d_u, d_h2=w.d_h2, d_x1=w.d_x1,
                    d_v=w.d_v, d_h1=w.d_h1, d_rgb_pre=w.d_rgb_pre, d_dep_pre=w.d_dep_pre, part_nf=w.chain_parts["nf"],
                    part_n2=w.chain_parts["n2"], part_n1=w.lnp_seam["n1"], part_dep=w.lnp_seam["dep"],
                    d_caq=gl["caq"], d_t1_res=gl["t2pre"], wq=wi0[:H], t1_pre=c["t1_pre"], m1d=c["m1"], r1d=c["r1"],
                    g1d=a.p(pl + "norm1.weight"), drop_d1=dmf("d1_0"), w_out=a.p(pl + "self_attn.out_proj.weight"),
                    sa_qkv=c["sa_qkv"], p_sa=c["p_sa"], drop_sa=dmf("sa_p0"), w_in=a.p(pl + "self_attn.in_proj_weight"),
                    caqin=gl["caqin"], t1pre_out=gl["t1pre"], sap=gl["sap"], sao=gl["sao"], saqkv=gl["saqkv"], sain=gl["sain"],
                    part_d1=w.chain_parts["d1"], drop_scale=dsc, N=N, S=S, K=K, H=H, add_xres=1, B=B, Q=Q, heads=heads,
                    planes={k: pls.ptr(k) for k in pls.keys if k.endswith("_t") and "dec_" not in k} if pls is not None else None)
            if er:
                self._erank_join()
                self._erank_backward(w, ws, dst=w.d_extra)
            w.tables[key].launch()
        elif st.get("paired"):
            # the query-side branch (cross-attention query projection, norm1, self-attention: parameter gradients only
            # for a one-layer decoder) and the memory-side chain into the fuser are independent: their GEMMs share launches
            c, gl, pl = w.layers[0], w.glayers[0], "transformer.decoder.layers.0."
            split3 = bool(self.split_k4h and H < 256)
            key = ("bwd_pairs", bool(tail), split3)
            if key not in w.tables:
                wi0 = a.p(pl + "multihead_attn.in_proj_weight")
                t = 1 if H < 256 else 2
                first = [dict(a=gl["caq"], b=wi0[:H], c=gl["caqin"]), dict(a=gl["cakv"], b=wi0[H:], c=w.d_fused)]
                if tail:                       # the segmentation head's input gradient (see the tail launch above)
                    first.append(dict(a=w.d_seg, b=a.p("fc_seg.weight"), c=w.d_fused2))
                # W_proj . W_v (parameters only): with it the two chained GEMMs at the end of the fuser's backward,
                # d_v = swap(d_x1 . W_proj) and d_h1 = d_v . W_v = swap(d_x1 . (W_proj . W_v)), share one launch
                if not hasattr(w, "wc"):        # (one buffer whatever variant of the groups is built later)
                    w.wc = torch.empty(H, H, dtype=torch.float32, device=self.device)
                first.append(dict(a=a.p(pre + "attn.proj.weight"), b=a.p(pre + "attn.qkv.weight")[2 * H:], c=w.wc))
                w.tables[("bwd_vh1",)] = ops.GemmGroup(GEMM_NN, [
                    dict(a=w.d_x1, b=a.p(pre + "attn.proj.weight"), c=w.d_v, c_row_xor=1),
                    dict(a=w.d_x1, b=w.wc, c=w.d_h1, c_row_xor=1)], tile=t)
                # ride_attention: the query self-attention's backward core rides in the fuser's norm2-backward launch, so
                # its input-projection gradient moves from the third group to the last one of the chain
                w.tables[("bwd_ride",)] = (
                    ops.GemmGroup(GEMM_NN, [dict(a=w.d_u, b=a.p(pre + "mlp.mlp.0.weight"), c=w.d_h2)], tile=t),
                    ops.GemmGroup(GEMM_NN, [
                        dict(a=w.d_x1, b=a.p(pre + "attn.proj.weight"), c=w.d_v, c_row_xor=1),
                        dict(a=w.d_x1, b=w.wc, c=w.d_h1, c_row_xor=1),
                        dict(a=gl["saqkv"], b=a.p(pl + "self_attn.in_proj_weight"), c=gl["sain"])], tile=t))
                w.tables[key] = (
                    ops.GemmGroup(GEMM_NN, first, tile=t),
                    ops.GemmGroup(GEMM_NN, [dict(a=gl["sap"], b=a.p(pl + "self_attn.out_proj.weight"), c=gl["sao"]),
                                            dict(a=w.d_x3, b=a.p(pre + "mlp.mlp.2.weight"), c=w.d_u, aux=w.u, mul=2)], tile=t),
                    ops.GemmGroup(GEMM_NN, [dict(a=gl["saqkv"], b=a.p(pl + "self_attn.in_proj_weight"), c=gl["sain"])] + (
                        [dict(a=w.d_u[:, :2 * H], b=a.p(pre + "mlp.mlp.0.weight")[:2 * H], c=w.d_h2),
                         dict(a=w.d_u[:, 2 * H:], b=a.p(pre + "mlp.mlp.0.weight")[2 * H:], c=w.d_h2b)]
                        if split3 else
                        [dict(a=w.d_u, b=a.p(pre + "mlp.mlp.0.weight"), c=w.d_h2)]), tile=t))
            gb1, gb2, gb3 = w.tables[key]
            gb1.launch()
            if st.get("erank"):
                self._erank_join()
                self._erank_backward(w, ws)
            def lnj(site, dy, x, mean, rstd, gname, bname, dx, **kw):
                return dict(dy=dy, x=x, mean=mean, rstd=rstd, gamma=a.p(gname), beta=a.p(bname), dx=dx, dgamma=a.g(gname),
                            dbeta=a.g(bname), partial=w.lnp[site], **kw)
            ops.layernorm_bwd_multi([                  # decoder norm1 and the fuser's final norm: one launch
                lnj("d1_0", gl["caqin"], c["t1_pre"], c["m1"], c["r1"], pl + "norm1.weight", pl + "norm1.bias", gl["t1pre"],
                    dy2=gl["t2pre"], dx2=gl["sap"], drop_mask=dm("d1_0", BQ, H), drop_scale=dsc),
                lnj("nf", w.d_fused, w.x3, w.mf, w.rf, "fuser.norm.weight", "fuser.norm.bias", w.d_x3, pair_in=True,
                    dy2=w.d_fused2)])
            gb2.launch()
            ride_bwd = bool(self.ride_attention_bwd and H <= 128 and ops.gemm_ln_mha_supported(heads, Q, Q, dh))
            if ride_bwd:
                w.tables[("bwd_ride",)][0].launch()                  # d_u -> d_h2 alone; the attention core follows below
            else:
                ops.mha_core_bwd(c["sa_qkv"][:, :H], c["sa_qkv"][:, H:2 * H], c["sa_qkv"][:, 2 * H:], c["p_sa"], gl["sao"],
                                 gl["saqkv"][:, :H], gl["saqkv"][:, H:2 * H], gl["saqkv"][:, 2 * H:], B, heads, Q, Q, dh,
                                 drop_mask=dmf("sa_p0"), drop_scale=dsc)
                gb3.launch()
        else:
            if st.get("erank"):
                self._erank_join()
                self._erank_backward(w, ws)
            ln_bwd("nf", w.d_fused, w.x3, w.mf, w.rf, "fuser.norm.weight", "fuser.norm.bias", w.d_x3, pair_in=True,
                   dy2=w.d_fused2)
            ops.gemm(GEMM_NN, w.d_x3, a.p(pre + "mlp.mlp.2.weight"), w.d_u, aux=w.u, mul=2, ws=ws)
            ops.gemm(GEMM_NN, w.d_u, a.p(pre + "mlp.mlp.0.weight"), w.d_h2, ws=ws)
        ride_bwd = bool(st.get("paired") and not chain and self.ride_attention_bwd and H <= 128 and
                        ops.gemm_ln_mha_supported(heads, Q, Q, dh))
        if chain:
            pass
        elif ride_bwd:
            c0, gl0 = w.layers[0], w.glayers[0]
            ops.layernorm_bwd_multi(
                [dict(dy=w.d_h2, x=w.x1, mean=w.m2, rstd=w.r2, gamma=a.p(pre + "norm2.weight"), beta=a.p(pre + "norm2.bias"),
                      dx=w.d_x1, dgamma=a.g(pre + "norm2.weight"), dbeta=a.g(pre + "norm2.bias"), partial=w.lnp["n2"],
                      add1=w.d_x3)],
                mha=dict(q=c0["sa_qkv"][:, :H], k=c0["sa_qkv"][:, H:2 * H], v=c0["sa_qkv"][:, 2 * H:], probs=c0["p_sa"],
                         d_o=gl0["sao"], dq=gl0["saqkv"][:, :H], dk=gl0["saqkv"][:, H:2 * H], dv=gl0["saqkv"][:, 2 * H:], B=B,
                         heads=heads, Lq=Q, Lk=Q, dh=dh, drop_mask=dmf("sa_p0"), drop_scale=dsc))
            w.tables[("bwd_ride",)][1].launch()
        else:
            ln_bwd("n2", w.d_h2, w.x1, w.m2, w.r2, pre + "norm2.weight", pre + "norm2.bias", w.d_x1, add1=w.d_x3,
                   dy2=w.d_h2b if (st.get("paired") and self.split_k4h and H < 256) else None)
        if ride_bwd or chain:
            pass
        elif st.get("paired"):
            w.tables[("bwd_vh1",)].launch()
        else:
            ops.gemm(GEMM_NN, w.d_x1, a.p(pre + "attn.proj.weight"), w.d_v, c_row_xor=1, ws=ws)     # un-swap
            ops.gemm(GEMM_NN, w.d_v, a.p(pre + "attn.qkv.weight")[2 * H:], w.d_h1, ws=ws)
        mask = st["mask"]
        if chain:
            pass
        elif st["seam"]:                    # norm1 backward + exchange backward + depth LayerNorm backward: one launch
            ops.embed_fuse_bwd(w.d_h1, w.x0, w.m1, w.r1, a.p(pre + "norm1.weight"), w.d_x1, w.d_x3, dmf("x0"), dsc, mask[0],
                               mask[1], w.rgb, w.dep_pre, w.mean_d, w.rstd_d, a.p("depth_layernorm.weight"),
                               a.p("depth_layernorm.bias"), w.d_rgb_pre, w.d_dep_pre, w.lnp_seam["n1"], w.lnp_seam["dep"])
        elif self.bn:
            t = w.bn_terms
            ops.bn_blend_bwd(w.d_h1, w.x0, w.m1, w.r1, a.p(pre + "norm1.weight"), w.d_x1, dmf("x0"), dsc, w.rgb, w.dep,
                             w.bn_mean, w.bn_rstd, a.p("fuser.bn_rgb.weight"), a.p("fuser.bn_rgb.bias"),
                             a.p("fuser.bn_depth.weight"), a.p("fuser.bn_depth.bias"), a.p("fuser.alpha").view(-1), mask[0],
                             mask[1], t[0], t[1], t[2], t[3], t[4], w.lnp_seam["n1"])
            w.bn_sums.launch()                 # column sums -> d gamma / d beta of both BatchNorms, d alpha
            if st["bn_training"] and self.bn_sync is not None:
                s4 = self.bn_sync.backward_sums(self, w)       # the sums over the global batch
                sums = (s4[0], s4[1], s4[2], s4[3])
            else:
                sums = (a.g("fuser.bn_rgb.weight"), a.g("fuser.bn_rgb.bias"), a.g("fuser.bn_depth.weight"),
                        a.g("fuser.bn_depth.bias"))
            ops.bn_bwd_apply(w.rgb, w.dep, w.bn_mean, w.bn_rstd, a.p("fuser.bn_rgb.weight"), a.p("fuser.bn_depth.weight"),
                             t[0], t[2], sums[0], sums[1], sums[2], sums[3], w.d_rgb_pre, w.d_dep, st["bn_training"])
            ln_bwd("dep", w.d_dep, w.dep_pre, w.mean_d, w.rstd_d, "depth_layernorm.weight", "depth_layernorm.bias",
                   w.d_dep_pre, relu=True)
        else:
            ln_bwd("n1", w.d_h1, w.x0, w.m1, w.r1, pre + "norm1.weight", pre + "norm1.bias", w.d_x0, add1=w.d_x1,
                   add2=w.d_x3)
            ops.token_exchange_bwd(w.d_x0, w.rgb, mask[0], mask[1], w.d_rgb_pre, w.d_dep, drop_mask=dmf("x0"),
                                   drop_scale=dsc)
            # ---- embeddings
            ln_bwd("dep", w.d_dep, w.dep_pre, w.mean_d, w.rstd_d, "depth_layernorm.weight", "depth_layernorm.bias",
                   w.d_dep_pre, relu=True)
        if not joined:
            main.wait_stream(s2)
        if self._overlap_tail_now:
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self._param_tail(w, st, chain)
            self._tail_pending = True
        else:
            self._param_tail(w, st, chain)

    def _param_tail(self, w, st, chain):
        """Everything of the backward that only feeds parameter gradients: 2 launches + the broadcast-parameter sums."""
        a, Q = self.arena, self.Q
        tname = "chain" if chain else ("bn" if self.bn else ("seam" if st["seam"] else "plain"))
        both = self._wgrad_with_sums(w, tname) if (self.fold_rowsums and self.L == 1) else None
        wg = both if both is not None else w.wgrad_group
        wg.set_b(w.rgb_wgrad_idx, st["x_rgb"])
        wg.launch()
        tail = w.tail_groups.get(tname)
        assert not chain or both is not None or tail is not None, "chain backward: its partial layout needs the row-sum tail"
        if both is not None:               # (the sums below rode in the weight-gradient launch)
            pass
        elif tail is not None:             # pos_embedding (:190), depth_projection.bias, query_embed (top layer) and
            tail.launch()                  # every LayerNorm parameter gradient: one launch
        else:
            assert not self.bn, "BN-blend variant: LayerNorm gradient slots must be adjacent in the arena"
            (w.ln_group_seam if st["seam"] else w.ln_group).launch()
            w.rowsum_group.launch()
        g_qe = a.g("query_embed.weight")
        for l in reversed(range(self.L - 1)):                       # stacked decoders: remaining layers accumulate
            gl = w.glayers[l]
            ops.rowmod_sum(gl["caqin"], Q, g_qe, accumulate=True)
            ops.rowmod_sum(gl["sain"], Q, g_qe, accumulate=True)

    # ------------------------------------------------------------------------------------------------------
    def set_lr(self, lr):
        if self._lr_host != float(lr):          # lr lives in device memory so a captured graph sees scheduler updates
            self.lr_t.fill_(float(lr))
            self._lr_host = float(lr)

    def adamw(self, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, tick_dropout=False, ticked=False,
              skip_depth=False, prefill_dropout=False, before_flat=None):
        """One fused launch over the live prefix of the arena (main_darai.py:135; train_proposed_depth.py:215).
        before_flat: callable run between the pixel-sharded weight's update (which needs no exchanged gradient, so it goes
        first) and the flat launch over the replicated parameters (which does) -- the place to join their all-reduce.
        ticked: losses(tick=True) already advanced the counters in this step.
        skip_depth: backward(fused_adamw=...) already updated depth_projection.weight.
        prefill_dropout: the same launch also fills the dropout pool of the step's workspace with the NEXT step's masks
        (valid while the next forward uses the same shape and the dropout offset is not advanced again)."""
        a = self.arena
        self.set_lr(lr)
        kw = dict(beta1=betas[0], beta2=betas[1], eps=eps, weight_decay=weight_decay, grad_scale=grad_scale)
        n0 = 0
        if self._tail_pending and ticked and self.tp is None and not skip_depth and before_flat is None:
            # the branch that produced the small bucket's gradients goes on to update it; this stream keeps the one weight
            # whose gradient it has just written (AdamW is element-wise: the cut changes no bit)
            n0 = a.bucket_small[1]
            with torch.cuda.stream(self.side):
                ops.adamw_flat(a.params[:n0], a.grads[:n0], a.exp_avg[:n0], a.exp_avg_sq[:n0], self.lr_t, self.step_t, **kw)
        else:
            self._join_tail()
        if not ticked:
            ops.tick(self.step_t, self.drop_offset if tick_dropout else None)
        n = a.n_live if (self.tp is None and not skip_depth) else a.bucket_small[1]
        st = self.last
        depth_cols_first = before_flat is not None and self.tp is not None and not skip_depth
        if depth_cols_first:
            t = self.tp
            ops.adamw_2d(t.w, t.g, t.m, t.v, self.lr_t, self.step_t, beta1=betas[0], beta2=betas[1], eps=eps,
                         weight_decay=weight_decay, grad_scale=grad_scale)
        if before_flat is not None:
            before_flat()
        pending, self._loss_pending = getattr(self, "_loss_pending", None), None
        if prefill_dropout and st is not None and st["drop"] and (ticked or tick_dropout):
            ops.adamw_flat_dropout(a.params[n0:n], a.grads[n0:n], a.exp_avg[n0:n], a.exp_avg_sq[n0:n], self.lr_t, self.step_t,
                                   st["w"].drop_pool, DROP_P, self.drop_seed, self.drop_offset, loss_fin=pending, **kw)
            self._drop_ready = st["w"]
            pending = None
        else:
            ops.adamw_flat(a.params[n0:n], a.grads[n0:n], a.exp_avg[n0:n], a.exp_avg_sq[n0:n], self.lr_t, self.step_t,
                           loss_fin=pending, **kw)
        self._join_tail()
        if self.tp is not None and not skip_depth and not depth_cols_first:
            t = self.tp                             # depth_projection.weight: only this rank's pixel columns are live
            ops.adamw_2d(t.w, t.g, t.m, t.v, self.lr_t, self.step_t, beta1=betas[0], beta2=betas[1], eps=eps,
                         weight_decay=weight_decay, grad_scale=grad_scale)

    def train_step(self, feats, depth, past_label, target_dur, target, lr, weight_decay, training=True):
        """forward + losses + backward + AdamW, all enqueued, no host sync.  Returns (loss[4], counts[4]) on device."""
        keep, self.defer_tail = self.defer_tail, True       # forward -> losses -> backward back to back: one tail launch
        keep_r, self.defer_loss_reduce = self.defer_loss_reduce, True     # loss statistics reduced in the AdamW launch
        try:
            self.forward(feats, depth, past_label, "train", training)
            loss, counts = self.losses(past_label, target, target_dur, tick=True)
        finally:
            self.defer_tail, self.defer_loss_reduce = keep, keep_r
        # depth_projection.weight updated inside its weight-gradient kernel where that pays (depth_adamw_fusable: the tiled
        # bf16x3 kernel of the wide / long shapes; measured neutral on the headline shape's panel kernel, which keeps the
        # plain sequence)
        fuse = self.depth_adamw_fusable()
        self.backward(fused_adamw=dict(lr=lr, weight_decay=weight_decay) if fuse else None, adamw_next=True)
        self.adamw(lr, weight_decay, ticked=True, prefill_dropout=True, skip_depth=fuse)
        return loss, counts
