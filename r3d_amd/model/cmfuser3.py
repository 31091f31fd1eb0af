"""Build-defined THREE-modality SA-Fuser (BASELINE.json configs[4]: "Synthetic 3-modality (RGB+Depth+Gaze) fusion").

The reference's `CMFuser` is structurally two-token: `forward` hard-codes the keys 'rgb' / 'depth' and a 2 x 2 mask
(model/futr_safuser_tokenfusion.py:74-81).  SURVEY.md 8(d) allows an M = 3 extension checked against the build's own CPU
restatement (oracle/futr_oracle.py: cm_fuser_m) -- **build-defined, parity unpinned**: no reference output exists for it.
The extension keeps every line of the original that generalises:

    token_fusion (:33-66)   per-modality scores exactly as in the reference (train: |d mean / dx| = a constant, so the
                            selection is the reference's tie rule; eval: mean |x| per channel), k = C // 4 lowest-score
                            channels of modality m are overwritten with the same channels of the NEXT modality
                            (m -> (m + 1) mod 3: the cyclic form of "rgb takes depth's, depth takes rgb's");
    mask (:68-72)           generate_cross_attention_mask(3): -inf on the diagonal, so every token attends to the two OTHER
                            modality tokens -- a real softmax over two logits, Q / K projections now receive gradient
                            (for M = 2 this is the swap the two-modality path computes in closed form);
    Block, + x_res, norm, mean over the tokens (:83-94, model/extras/transformerblock.py:118-135) unchanged.

Same parameter names / shapes as the reference's CMFuser (blocks.0.{norm1, attn.qkv, attn.proj, norm2, mlp.mlp.0, mlp.mlp.2},
norm; dead: modality_token, projection, fusion_conv), so a two-modality checkpoint's fuser loads into it.

Everything runs in HIP through the C ABI: the library's GEMM / LayerNorm entry points plus csrc/fuser3.hip (3-way exchange,
3-token attention core, mean over token triples, and their adjoints); torch only owns the buffers and the autograd node.
"""
import torch
import torch.nn as nn

from .. import ops
from .._lib import GEMM_NT, GEMM_NN, GEMM_TN
from .futr_safuser_tokenfusion import _Block

DROP_P = 0.1            # embd_drop (:26)


class _Fuser3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xa, xb, xc, mask, drop, heads, g1, b1, wqkv, wproj, bproj, g2, b2, w1, bb1, w2, bb2, gF, bF):
        dev, N, C = xa.device, xa.shape[0], xa.shape[1]
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)     # noqa: E731
        R = 3 * N
        ws = ops.GemmWorkspace(dev)
        dsc = 1.0 / (1.0 - DROP_P)
        x0, h1, m1, r1 = f(R, C), f(R, C), f(R), f(R)
        ops.token_exchange3_fwd(xa, xb, xc, mask, x0, drop_mask=drop, drop_scale=dsc)
        ops.layernorm_fwd(x0, g1, b1, h1, m1, r1)
        qkv, probs, att = f(R, 3 * C), f(N * heads * 6), f(R, C)
        ops.gemm(GEMM_NT, h1, wqkv, qkv, ws=ws)                                     # qkv_bias=False (:19,22)
        ops.attn3_fwd(qkv, probs, att, heads)
        x1, h2, m2, r2 = f(R, C), f(R, C), f(R), f(R)
        ops.gemm(GEMM_NT, att, wproj, x1, bias=bproj, res1=x0, ws=ws)              # x + attn (transformerblock.py:131)
        ops.layernorm_fwd(x1, g2, b2, h2, m2, r2)
        u, f1, x3 = f(R, 4 * C), f(R, 4 * C), f(R, C)
        ops.gemm(GEMM_NT, h2, w1, f1, bias=bb1, act=2, pre_out=u, ws=ws)            # GELU (:80,86)
        ops.gemm(GEMM_NT, f1, w2, x3, bias=bb2, res1=x1, res2=x0, ws=ws)           # + mlp, + x_res (:92)
        y, mf, rf, fused = f(R, C), f(R), f(R), f(N, C)
        ops.layernorm_fwd(x3, gF, bF, y, mf, rf)
        ops.triple_mean_fwd(y, fused)
        ctx.heads, ctx.dsc = heads, dsc
        ctx.save_for_backward(mask, drop, x0, h1, m1, r1, qkv, att, x1, h2, m2, r2, u, f1, x3, mf, rf, g1, b1, wqkv, wproj, g2, b2,
                              w1, w2, gF, bF)
        return fused

    @staticmethod
    def backward(ctx, d_fused):
        (mask, drop, x0, h1, m1, r1, qkv, att, x1, h2, m2, r2, u, f1, x3, mf, rf, g1, b1, wqkv, wproj, g2, b2, w1, w2, gF,
         bF) = ctx.saved_tensors
        dev = x0.device
        R, C = x0.shape
        N = R // 3
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)     # noqa: E731
        ws = ops.GemmWorkspace(dev)
        d_fused = d_fused.contiguous().float()
        dy = f(R, C)
        ops.triple_mean_bwd(d_fused, dy)
        d_x3, dgF, dbF = f(R, C), f(C), f(C)
        ops.layernorm_bwd(dy, x3, mf, rf, gF, bF, d_x3, dgF, dbF, ws=ws)
        d_u, dw2, dbb2 = f(R, 4 * C), torch.empty_like(w2), f(C)
        ops.gemm(GEMM_TN, d_x3, f1, dw2, bias_grad=dbb2, ws=ws)
        ops.gemm(GEMM_NN, d_x3, w2, d_u, aux=u, mul=2, ws=ws)                       # * GELU'(u)
        d_h2, dw1, dbb1 = f(R, C), torch.empty_like(w1), f(4 * C)
        ops.gemm(GEMM_TN, d_u, h2, dw1, bias_grad=dbb1, ws=ws)
        ops.gemm(GEMM_NN, d_u, w1, d_h2, ws=ws)
        d_x1, dg2, db2 = f(R, C), f(C), f(C)
        ops.layernorm_bwd(d_h2, x1, m2, r2, g2, b2, d_x1, dg2, db2, add1=d_x3, ws=ws)
        d_att, dwp, dbp = f(R, C), torch.empty_like(wproj), f(C)
        ops.gemm(GEMM_TN, d_x1, att, dwp, bias_grad=dbp, ws=ws)
        ops.gemm(GEMM_NN, d_x1, wproj, d_att, ws=ws)
        d_qkv = f(R, 3 * C)
        ops.attn3_bwd(qkv, d_att, d_qkv, ctx.heads)
        d_h1, dwqkv = f(R, C), torch.empty_like(wqkv)
        ops.gemm(GEMM_TN, d_qkv, h1, dwqkv, ws=ws)
        ops.gemm(GEMM_NN, d_qkv, wqkv, d_h1, ws=ws)
        d_x0, dg1, db1 = f(R, C), f(C), f(C)
        ops.layernorm_bwd(d_h1, x0, m1, r1, g1, b1, d_x0, dg1, db1, add1=d_x1, add2=d_x3, ws=ws)    # residual + x_res
        da, db, dc = f(N, C), f(N, C), f(N, C)
        ops.token_exchange3_bwd(d_x0, mask, da, db, dc, drop_mask=drop, drop_scale=ctx.dsc)
        return (da, db, dc, None, None, None, dg1, db1, dwqkv, dwp, dbp, dg2, db2, dw1, dbb1, dw2, dbb2, dgF, dbF)


class CMFuser3(nn.Module):
    """forward({'rgb': x, 'depth': x, <third key>: x}, mode) with x [B, T, C] on the device -> fused [B, T, C]
    (three modalities in the dict's order; the reference's two-key call is model.futr_safuser_tokenfusion.CMFuser)."""

    def __init__(self, dim, depth=1, num_heads=4, mlp_ratio=4., qkv_bias=False):
        super().__init__()
        assert depth == 1 and not qkv_bias and mlp_ratio == 4.
        self.blocks = nn.ModuleList([_Block(dim, num_heads)])
        self.norm = nn.LayerNorm(dim)
        self.embd_drop = nn.Dropout(DROP_P)
        self.modality_token = nn.Parameter(torch.randn(1, 1, 1, dim))          # dead, as in the reference (:28-31)
        self.projection = nn.Linear(dim, dim)
        self.fusion_conv = nn.Conv2d(in_channels=2, out_channels=1, kernel_size=1)
        self.dim, self.num_heads = dim, num_heads
        self.drop_seed, self._drop_calls = 0x5EED3, 0
        self.last_idx = None

    def select(self, xs, mode):
        """token_fusion's scores and top-k (:40-54) per modality -> (idx [3, k] int64, mask [3, C] float)."""
        C = self.dim
        dev = xs[0].device
        k = C // 4
        idx = torch.empty(3, k, dtype=torch.int64, device=dev)
        mask = torch.empty(3, C, dtype=torch.float32, device=dev)
        if mode == "train":
            n = xs[0].shape[0]
            sc = torch.full((3, C), 1.0 / (n * C), dtype=torch.float32, device=dev)      # |d mean / dx| averaged (:40-45)
            ops.token_select(k, idx, mask, score_f=sc)
        else:
            sums = torch.empty(3, C, dtype=torch.float64, device=dev)
            for m, x in enumerate(xs):
                ops.colabssum(x, sums[m])
            ops.token_select(k, idx, mask, score_sum=sums, count=float(xs[0].shape[0]))
        return idx, mask

    def forward(self, modal_feats, mode="train"):
        xs = list(modal_feats.values())
        assert len(xs) == 3, "CMFuser3 fuses exactly three modalities"
        B, T, C = xs[0].shape
        assert C == self.dim and xs[0].is_cuda, "the HIP fuser needs device tensors (there is no CPU path)"
        flat = [x.reshape(B * T, C).contiguous().float() for x in xs]
        idx, mask = self.select(flat, mode)
        self.last_idx = idx
        drop = None
        if self.training:
            drop = torch.empty(3 * B * T * C, dtype=torch.uint8, device=xs[0].device)
            off = torch.full((1,), self._drop_calls, dtype=torch.int64, device=xs[0].device)
            ops.dropout_mask(drop, DROP_P, self.drop_seed, off)
            self._drop_calls += 1
            drop = drop.view(3 * B * T, C)
        blk = self.blocks[0]
        wqkv = blk.attn.qkv.weight
        fused = _Fuser3Fn.apply(flat[0], flat[1], flat[2], mask, drop, self.num_heads, blk.norm1.weight, blk.norm1.bias, wqkv,
                                blk.attn.proj.weight, blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias,
                                blk.mlp.mlp[0].weight, blk.mlp.mlp[0].bias, blk.mlp.mlp[2].weight, blk.mlp.mlp[2].bias,
                                self.norm.weight, self.norm.bias)
        return fused.view(B, T, C)
