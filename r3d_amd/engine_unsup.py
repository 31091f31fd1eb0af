"""Host-side sequencing of the depth-as-query model (reference model/futr_unsupervised_depth.py) on one MI355X.

The second model file BASELINE.json's north star names for the per-modality encoder embeddings: the RGB embedding
(+ sinusoidal PositionalEncoding, model/extras/position.py:15-35) is the decoder's MEMORY (the DETR encoder is
constructed but bypassed, transformer.py:77-78) and the depth embedding 160*120 -> H + LayerNorm + ReLU (+ the same
encoding) is its per-clip QUERY -- S queries per clip, different for every clip, so unlike the token-fusion model the
query-side self-attention depends on the data and its gradient flows back into the depth projection.  The S decoder
outputs are average-pooled to n_query rows before the anticipation heads (futr_unsupervised_depth.py:134).

Same flat arenas, workspaces, C ABI and method surface (forward / losses / backward / adamw / train_step) as
engine.FusionEngine, so r3d_amd.train_proposed_depth.train() drives either model; every product is a launch of
libr3d_hip.so (GEMMs with prologues / epilogues, the attention core, LayerNorm, the loss kernel, fused AdamW, plus the
positional-encoding and pooling kernels of csrc/posenc.hip).  This variant is composed launch by launch (no paired /
seam / tail fusions): it is a coverage row of SURVEY.md 8(a) A2/A3, not the headline workload.
"""
import torch

from . import ops
from ._lib import GEMM_NT, GEMM_NN, GEMM_TN
from .engine import ParamArena, DROP_P, EXCLUDE_CLASS_IDX

LIVE_PREFIXES = ("input_embed.", "depth_projection.", "depth_layernorm.", "pos_embedding", "transformer.decoder.",
                 "fc_seg.", "fc.", "fc_len.", "query_embed.")     # (query_embed: the label-query variant, model/futr_proposed.py)


class _Arena(ParamArena):
    def __init__(self, named_params, device):
        # ParamArena's rule set is the token-fusion model's; this model has no query_embed / fuser, so the shared prefixes
        # select exactly the parameters that receive a gradient here (checked against the reference fixture's live set)
        super().__init__(named_params, device)
        self.is_live = lambda n: n.startswith(LIVE_PREFIXES)


class _Shape:
    def __init__(self, eng, B, S, train):
        dev, H, Q, K, L, heads = eng.device, eng.H, eng.Q, eng.K, eng.L, eng.heads
        N, BQ = B * S, B * Q
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)     # noqa: E731
        self.B, self.S, self.N, self.BQ = B, S, N, BQ
        self.rgb, self.mem = f(N, H), f(N, H)
        self.dep_pre, self.dep, self.qpos = f(N, H), f(N, H), f(N, H)
        self.mean_d, self.rstd_d = f(N), f(N)
        self.tgt0 = torch.zeros(N, H, dtype=torch.float32, device=dev)
        self.layers = [dict(sa_qkv=f(N, 3 * H), sa_o=f(N, H), p_sa=f(B, heads, S, S), t1_pre=f(N, H), t1=f(N, H), m1=f(N),
                            r1=f(N), caq=f(N, H), cakv=f(N, 2 * H), ca_o=f(N, H), p_ca=f(B, heads, S, S), t2_pre=f(N, H),
                            t2=f(N, H), m2=f(N), r2=f(N), ff1=f(N, 4 * H), t3_pre=f(N, H), t3=f(N, H), m3=f(N), r3=f(N))
                       for _ in range(L)]
        self.tgtF, self.mF, self.rF = f(N, H), f(N), f(N)
        self.pooled = f(BQ, H)
        self.actdur = f(BQ, K + 1)
        self.seg = f(N, eng.Kseg)
        self.loss = f(4)
        self.loss_ws = torch.zeros(ops.losses_ws_floats(B, S, Q), dtype=torch.float32, device=dev)
        self.counts = torch.zeros(4, dtype=torch.int64, device=dev)
        if train:
            self.d_actdur, self.d_seg = torch.zeros(BQ, K + 1, dtype=torch.float32, device=dev), f(N, eng.Kseg)
            self.d_pooled, self.d_tgtF, self.d_t = f(BQ, H), f(N, H), f(N, H)
            self.glayers = [dict(t3pre=f(N, H), ff2=f(N, H), ff1=f(N, 4 * H), t2=f(N, H), t2pre=f(N, H), cap=f(N, H),
                                 cao=f(N, H), caq=f(N, H), cakv=f(N, 2 * H), caqin=f(N, H), t1pre=f(N, H), sap=f(N, H),
                                 sao=f(N, H), saqkv=f(N, 3 * H), sain=f(N, H)) for _ in range(L)]
            self.d_mp, self.d_mem, self.d_qpos, self.d_qtmp = f(N, H), f(N, H), f(N, H), f(N, H)
            self.d_rgb_pre, self.d_dep, self.d_dep_pre = f(N, H), f(N, H), f(N, H)
            sizes = dict(pe_rgb=N * H, pe_dep=N * H)
            for l in range(L):
                sizes.update({f"sa_p{l}": B * heads * S * S, f"ca_p{l}": B * heads * S * S, f"d1_{l}": N * H,
                              f"d2_{l}": N * H, f"d3_{l}": N * H, f"ff_{l}": N * 4 * H})
            tot = sum((v + 15) // 16 * 16 for v in sizes.values())
            self.drop_pool = torch.ones(tot, dtype=torch.uint8, device=dev)
            self.drop, o = {}, 0
            for k, v in sizes.items():
                self.drop[k] = self.drop_pool[o:o + v]
                o += (v + 15) // 16 * 16


class UnsupDepthEngine:
    def __init__(self, module, device):
        self.module = module
        self.device = torch.device(device)
        assert self.device.type == "cuda", "the HIP engine needs an MI355X device (there is no CPU path)"
        ops._lib.load()
        self.H, self.Q, self.K = module.hidden_dim, module.n_query, module.n_class
        self.heads, self.L = module.n_head, module.num_decoder_layers
        self.dh = self.H // self.heads
        self.pad_idx = module.src_pad_idx
        # label_query: model/futr_proposed.py -- the decoder query is nn.Embedding(label indices) + a sinusoidal table
        # (:103-106), the memory carries no positional encoding (:92-97) and fc_seg has n_class - 1 outputs (:38)
        self.label_query = not hasattr(module, "depth_projection")
        self.D = module.input_embed.in_features
        self.P = None if self.label_query else module.depth_projection.in_features
        self.Kseg = module.fc_seg.out_features
        assert self.H % 8 == 0 and self.H % self.heads == 0
        self.arena = _Arena(list(module.named_parameters()), self.device)
        self.pe = module.pos_enc.pos_table[0]                     # [3000, H] sinusoid buffer (position.py:19-27)
        self.pe_depth = None if self.label_query else module.pos_enc_depth.pos_table[0]
        self.pe_l3 = module.positional_embedding_l3.to(self.device).contiguous() if self.label_query else None
        self.ws = ops.GemmWorkspace(self.device)
        self.dropout_enabled = bool(getattr(module, "r3d_dropout_enabled", True))
        self.erank_weight = 0.0                # (the rank penalty is defined on the fuser's tokens; this model has no fuser)
        self.defer_tail = False
        self.shapes = {}
        self.drop_seed = 0x5EED
        self.drop_offset = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.step_t = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._lr_host = None
        self.dur_den = None
        self.grad_hook = None
        self.tp = None
        self._drop_ready = None
        self.last = None
        a, K, H = self.arena, self.K, self.H
        o_w = a.offsets["fc.weight"][0]
        self.w_head = a.params[o_w:o_w + (K + 1) * H].view(K + 1, H)
        self.gw_head = a.grads[o_w:o_w + (K + 1) * H].view(K + 1, H)
        o_b = a.offsets["fc.bias"][0]
        assert a.offsets["fc_len.weight"][0] == o_w + K * H and a.offsets["fc_len.bias"][0] == o_b + K
        self.b_head = a.params[o_b:o_b + K + 1]
        self.gb_head = a.grads[o_b:o_b + K + 1]

    def _shape(self, B, S, train):
        key = (B, S, bool(train))
        if key not in self.shapes:
            self.shapes[key] = _Shape(self, B, S, train)
        return self.shapes[key]

    @staticmethod
    def _dm2(m, rows, cols):
        return None if m is None else m.view(rows, cols)

    # ------------------------------------------------------------------------------------------------------
    def forward(self, feats, depth, labels, mode="train", training=False, need_grad=True):
        """feats [B,S,D] f32, depth [B,S,...] f32 (flattened to [N,P]), labels [B,S] int64 (train mode only).
        futr_unsupervised_depth.py:85-163.  Returns views: seg [B,S,K], action [B,Q,K], duration [B,Q]."""
        a, H, Q, K, heads, dh = self.arena, self.H, self.Q, self.K, self.heads, self.dh
        B, S = feats.shape[0], feats.shape[1]
        N, BQ = B * S, B * Q
        assert feats.is_cuda and depth.is_cuda and feats.dtype == torch.float32
        x_rgb = feats.reshape(N, -1)
        assert x_rgb.shape[1] == self.D and x_rgb.is_contiguous()
        if self.label_query:                                       # `depth` holds the label indices of the queries [B, S]
            assert depth.dtype == torch.int64 and depth.numel() == N and depth.is_contiguous()
            x_dep = depth.reshape(N)
        else:
            assert depth.dtype == torch.float32
            x_dep = depth.reshape(N, -1)
            assert x_dep.shape[1] == self.P and x_dep.is_contiguous(), (x_dep.shape, self.P)
        w = self._shape(B, S, need_grad)
        drop = training and need_grad and self.dropout_enabled
        if drop:
            ops.dropout_mask(w.drop_pool, DROP_P, self.drop_seed, self.drop_offset)
        dsc = 1.0 / (1.0 - DROP_P)
        dm = (lambda k: w.drop[k]) if drop else (lambda k: None)
        key_labels = None
        if mode == "train":                                     # get_pad_mask (:89,161-162) inside the attention kernel
            assert labels.dtype == torch.int64 and labels.is_cuda and labels.is_contiguous()
            key_labels = labels
        ws = self.ws
        # ---- RGB embedding: relu(x W^T + b) + pos_table, dropout (:93-99) -> decoder memory
        ops.gemm(GEMM_NT, x_rgb, a.p("input_embed.weight"), w.rgb, bias=a.p("input_embed.bias"), act=1, ws=ws)
        if self.label_query:
            w.mem = w.rgb                                          # (futr_proposed.py:92-97: no encoding on the memory)
            ops.embed_gather_fwd(a.p("query_embed.weight"), x_dep, self.pe_l3, S, w.qpos)      # (:103-106)
        else:
            ops.posenc_fwd(w.rgb, self.pe, S, w.mem, drop_mask=dm("pe_rgb"), drop_scale=dsc)
        # ---- depth embedding: relu(LN(x W^T + b)) + pos_table, dropout (:107-115) -> decoder query
        d = None if self.label_query else ops.gemm(GEMM_NT, x_dep, a.p("depth_projection.weight"), w.dep_pre, bias=a.p("depth_projection.bias"), ws=ws,
                     defer_reduce=True)                      # (split-K: raw slabs, the LayerNorm launch sums them + bias)
        if d is None:
            pass
        elif d.splitk > 1:
            ops.layernorm_fwd(ws.buf, a.p("depth_layernorm.weight"), a.p("depth_layernorm.bias"), w.dep, w.mean_d, w.rstd_d,
                              relu=True, nsplit=d.splitk, bias=a.p("depth_projection.bias"), pre_out=w.dep_pre, rows=N, H=H)
        else:
            ops.layernorm_fwd(w.dep_pre, a.p("depth_layernorm.weight"), a.p("depth_layernorm.bias"), w.dep, w.mean_d,
                              w.rstd_d, relu=True)
        if not self.label_query:
            ops.posenc_fwd(w.dep, self.pe_depth, S, w.qpos, drop_mask=dm("pe_dep"), drop_scale=dsc)
        # ---- segmentation head on the memory (:148; transformer.py:128 returns it untouched)
        ops.gemm(GEMM_NT, w.mem, a.p("fc_seg.weight"), w.seg, bias=a.p("fc_seg.bias"), ws=ws)
        # ---- decoder, post-norm (transformer.py:281-330): S queries per clip, query_pos = w.qpos (an activation)
        pos = a.p("pos_embedding")[0, :S]
        tgt = w.tgt0                                              # tgt = zeros_like(action_query) (:126)
        for l in range(self.L):
            c, pl = w.layers[l], f"transformer.decoder.layers.{l}."
            p = lambda n: a.p(pl + n)         # noqa: E731
            ops.gemm(GEMM_NT, tgt, p("self_attn.in_proj_weight"), c["sa_qkv"], a_add=w.qpos, a_add_mod=N,
                     bias=p("self_attn.in_proj_bias"), ws=ws)             # q = k = v = tgt + query_pos (:289)
            ops.mha_core_fwd(c["sa_qkv"][:, :H], c["sa_qkv"][:, H:2 * H], c["sa_qkv"][:, 2 * H:], c["p_sa"], c["sa_o"], B,
                             heads, S, S, dh, drop_mask=dm(f"sa_p{l}"), drop_scale=dsc)
            ops.gemm(GEMM_NT, c["sa_o"], p("self_attn.out_proj.weight"), c["t1_pre"], bias=p("self_attn.out_proj.bias"),
                     drop_mask=self._dm2(dm(f"d1_{l}"), N, H), drop_scale=dsc, res1=None if l == 0 else tgt, ws=ws)
            ops.layernorm_fwd(c["t1_pre"], p("norm1.weight"), p("norm1.bias"), c["t1"], c["m1"], c["r1"])
            wi, bi = p("multihead_attn.in_proj_weight"), p("multihead_attn.in_proj_bias")
            ops.gemm(GEMM_NT, c["t1"], wi[:H], c["caq"], a_add=w.qpos, a_add_mod=N, bias=bi[:H], ws=ws)
            ops.gemm(GEMM_NT, w.mem, wi[H:], c["cakv"], a_add=pos, a_add_mod=S, bias=bi[H:], ws=ws)   # k = v = memory + pos
            ops.mha_core_fwd(c["caq"], c["cakv"][:, :H], c["cakv"][:, H:], c["p_ca"], c["ca_o"], B, heads, S, S, dh,
                             key_labels=key_labels, pad_idx=self.pad_idx, drop_mask=dm(f"ca_p{l}"), drop_scale=dsc)
            ops.gemm(GEMM_NT, c["ca_o"], p("multihead_attn.out_proj.weight"), c["t2_pre"],
                     bias=p("multihead_attn.out_proj.bias"), drop_mask=self._dm2(dm(f"d2_{l}"), N, H), drop_scale=dsc,
                     res1=c["t1"], ws=ws)
            ops.layernorm_fwd(c["t2_pre"], p("norm2.weight"), p("norm2.bias"), c["t2"], c["m2"], c["r2"])
            ops.gemm(GEMM_NT, c["t2"], p("linear1.weight"), c["ff1"], bias=p("linear1.bias"), act=1,
                     drop_mask=self._dm2(dm(f"ff_{l}"), N, 4 * H), drop_scale=dsc, ws=ws)
            ops.gemm(GEMM_NT, c["ff1"], p("linear2.weight"), c["t3_pre"], bias=p("linear2.bias"),
                     drop_mask=self._dm2(dm(f"d3_{l}"), N, H), drop_scale=dsc, res1=c["t2"], ws=ws)
            ops.layernorm_fwd(c["t3_pre"], p("norm3.weight"), p("norm3.bias"), c["t3"], c["m3"], c["r3"])
            tgt = c["t3"]
        ops.layernorm_fwd(tgt, a.p("transformer.decoder.norm.weight"), a.p("transformer.decoder.norm.bias"), w.tgtF, w.mF,
                          w.rF)
        # ---- adaptive average pooling of the S outputs to n_query rows (:134) + anticipation heads (:140-144)
        ops.avgpool_rows_fwd(w.tgtF, w.pooled, B, S, Q)
        ops.gemm(GEMM_NT, w.pooled, self.w_head, w.actdur, bias=self.b_head, ws=ws)
        self.last = dict(w=w, x_rgb=x_rgb, x_dep=x_dep, drop=drop, mode=mode, tp=None)
        return dict(seg=w.seg.view(B, S, self.Kseg), action=w.actdur[:, :K].view(B, Q, K), duration=w.actdur[:, K].view(B, Q))

    # ------------------------------------------------------------------------------------------------------
    def losses(self, past_label, target, target_dur, with_grad=True, val_mode=False, tick=False):
        """The 3 losses + counters of train_proposed_depth.py:171-213 in one launch; fills d_seg / d_actdur."""
        w, K = self.last["w"], self.K
        if self.label_query:
            raise NotImplementedError("the fused loss kernel belongs to train_proposed_depth.py's composition; the label-query "
                                      "model's loop (train/train_unsupervised.py) is out of scope -- use the autograd bridge")
        ta = self.step_t if tick else None
        tb = self.drop_offset if (tick and self.last["drop"]) else None
        ops.losses_fwd_bwd(None if val_mode else w.seg, w.actdur[:, :K], w.actdur[:, K:], K + 1, past_label, target,
                           target_dur, w.B, w.S, self.Q, K, self.pad_idx, EXCLUDE_CLASS_IDX, w.loss, w.counts,
                           val_mode=val_mode, dur_den=self.dur_den,
                           d_seg=w.d_seg if with_grad else None, d_act=w.d_actdur[:, :K] if with_grad else None,
                           d_dur=w.d_actdur[:, K:] if with_grad else None, ld_ddur=K + 1, ws=w.loss_ws, tick_a=ta, tick_b=tb)
        return w.loss, w.counts

    # ------------------------------------------------------------------------------------------------------
    def backward(self, d_seg=None, d_actdur=None, fused_adamw=None, adamw_next=False):
        """Adjoint of forward(); gradients land in the grad arena (written, not accumulated).  (adamw_next: accepted for
        FusionEngine's signature; this engine has no parallel parameter-gradient branch.)"""
        assert fused_adamw is None
        st = self.last
        w, a, H, Q, K, heads, dh, ws = st["w"], self.arena, self.H, self.Q, self.K, self.heads, self.dh, self.ws
        B, S, N = w.B, w.S, w.N
        if d_seg is not None and d_seg.data_ptr() != w.d_seg.data_ptr():
            w.d_seg.copy_(d_seg)
        if d_actdur is not None and d_actdur.data_ptr() != w.d_actdur.data_ptr():
            w.d_actdur.copy_(d_actdur)
        drop = st["drop"]
        dsc = 1.0 / (1.0 - DROP_P)
        dm = (lambda k, r, c: w.drop[k].view(r, c)) if drop else (lambda k, r, c: None)
        dmf = (lambda k: w.drop[k]) if drop else (lambda k: None)
        pos = a.p("pos_embedding")[0, :S]

        def ln_bwd(dy, x, mean, rstd, gname, bname, dx, **kw):
            ops.layernorm_bwd(dy, x, mean, rstd, a.p(gname), a.p(bname), dx, a.g(gname), a.g(bname), ws=ws, **kw)

        def wgrad(dy, x, gw, gb, **kw):
            ops.gemm(GEMM_TN, dy, x, gw, bias_grad=gb, ws=ws, **kw)

        # ---- heads, pooling, decoder.norm
        wgrad(w.d_actdur, w.pooled, self.gw_head, self.gb_head)
        ops.gemm(GEMM_NN, w.d_actdur, self.w_head, w.d_pooled, ws=ws)
        ops.avgpool_rows_bwd(w.d_pooled, w.d_tgtF, B, S, Q)
        last = w.layers[-1]
        ln_bwd(w.d_tgtF, last["t3"], w.mF, w.rF, "transformer.decoder.norm.weight", "transformer.decoder.norm.bias", w.d_t)
        dy, dy2 = w.d_t, None
        for l in reversed(range(self.L)):
            c, gl, pl = w.layers[l], w.glayers[l], f"transformer.decoder.layers.{l}."
            p = lambda n: a.p(pl + n)         # noqa: E731
            g = lambda n: a.g(pl + n)         # noqa: E731
            tgt_in = w.tgt0 if l == 0 else w.layers[l - 1]["t3"]
            ln_bwd(dy, c["t3_pre"], c["m3"], c["r3"], pl + "norm3.weight", pl + "norm3.bias", gl["t3pre"], dy2=dy2,
                   dx2=gl["ff2"], drop_mask=dm(f"d3_{l}", N, H), drop_scale=dsc)
            wgrad(gl["ff2"], c["ff1"], g("linear2.weight"), g("linear2.bias"))
            ops.gemm(GEMM_NN, gl["ff2"], p("linear2.weight"), gl["ff1"], drop_mask=dm(f"ff_{l}", N, 4 * H), drop_scale=dsc,
                     aux=c["ff1"], mul=1, ws=ws)
            wgrad(gl["ff1"], c["t2"], g("linear1.weight"), g("linear1.bias"))
            ops.gemm(GEMM_NN, gl["ff1"], p("linear1.weight"), gl["t2"], res1=gl["t3pre"], ws=ws)
            ln_bwd(gl["t2"], c["t2_pre"], c["m2"], c["r2"], pl + "norm2.weight", pl + "norm2.bias", gl["t2pre"],
                   dx2=gl["cap"], drop_mask=dm(f"d2_{l}", N, H), drop_scale=dsc)
            wgrad(gl["cap"], c["ca_o"], g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias"))
            ops.gemm(GEMM_NN, gl["cap"], p("multihead_attn.out_proj.weight"), gl["cao"], ws=ws)
            ops.mha_core_bwd(c["caq"], c["cakv"][:, :H], c["cakv"][:, H:], c["p_ca"], gl["cao"], gl["caq"], gl["cakv"][:, :H],
                             gl["cakv"][:, H:], B, heads, S, S, dh, drop_mask=dmf(f"ca_p{l}"), drop_scale=dsc)
            wi = p("multihead_attn.in_proj_weight")
            gwi, gbi = g("multihead_attn.in_proj_weight"), g("multihead_attn.in_proj_bias")
            wgrad(gl["cakv"], w.mem, gwi[H:], gbi[H:], b_add=pos, b_add_mod=S)
            wgrad(gl["caq"], c["t1"], gwi[:H], gbi[:H], b_add=w.qpos, b_add_mod=N)
            ops.gemm(GEMM_NN, gl["cakv"], wi[H:], w.d_mp, accumulate=(l != self.L - 1), ws=ws)     # d (memory + pos)
            ops.gemm(GEMM_NN, gl["caq"], wi[:H], gl["caqin"], ws=ws)                              # d (t1 + query_pos)
            ln_bwd(gl["caqin"], c["t1_pre"], c["m1"], c["r1"], pl + "norm1.weight", pl + "norm1.bias", gl["t1pre"],
                   dy2=gl["t2pre"], dx2=gl["sap"], drop_mask=dm(f"d1_{l}", N, H), drop_scale=dsc)
            wgrad(gl["sap"], c["sa_o"], g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"))
            ops.gemm(GEMM_NN, gl["sap"], p("self_attn.out_proj.weight"), gl["sao"], ws=ws)
            ops.mha_core_bwd(c["sa_qkv"][:, :H], c["sa_qkv"][:, H:2 * H], c["sa_qkv"][:, 2 * H:], c["p_sa"], gl["sao"],
                             gl["saqkv"][:, :H], gl["saqkv"][:, H:2 * H], gl["saqkv"][:, 2 * H:], B, heads, S, S, dh,
                             drop_mask=dmf(f"sa_p{l}"), drop_scale=dsc)
            wgrad(gl["saqkv"], tgt_in, g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"), b_add=w.qpos, b_add_mod=N)
            ops.gemm(GEMM_NN, gl["saqkv"], p("self_attn.in_proj_weight"), gl["sain"], ws=ws)      # d (tgt_in + query_pos)
            # d query_pos += caqin + sain  (the query is an activation here: its gradient reaches the depth projection)
            if l == self.L - 1:
                ops.add_rowbcast(gl["caqin"], gl["sain"], N, w.d_qpos)
            else:
                ops.add_rowbcast(gl["caqin"], gl["sain"], N, w.d_qtmp)
                ops.add_rowbcast(w.d_qpos, w.d_qtmp, N, w.d_qpos)
            dy, dy2 = gl["sain"], gl["t1pre"]           # d t3 of layer l-1 = through the queries + the residual
        # ---- learned positional embedding (:104): column sums over the clips of d (memory + pos)
        ops.rowmod_sum(w.d_mp, S, a.g("pos_embedding")[0, :S])
        # ---- memory: decoder part + segmentation head part; through the encoding's dropout and the ReLU (:97-99)
        wgrad(w.d_seg, w.mem, a.g("fc_seg.weight"), a.g("fc_seg.bias"))
        ops.gemm(GEMM_NN, w.d_seg, a.p("fc_seg.weight"), w.d_mem, res1=w.d_mp, ws=ws)
        ops.posenc_bwd(w.d_mem, w.d_rgb_pre, drop_mask=None if self.label_query else dmf("pe_rgb"), drop_scale=dsc, gate=w.rgb)
        wgrad(w.d_rgb_pre, st["x_rgb"], a.g("input_embed.weight"), a.g("input_embed.bias"))
        if self.label_query:                                       # the lookup's adjoint (futr_proposed.py:103)
            ops.embed_gather_bwd(w.d_qpos, st["x_dep"], a.g("query_embed.weight"))
            return
        # ---- query: through the encoding's dropout, ReLU + LayerNorm (:110-115), into the depth projection (:109)
        ops.posenc_bwd(w.d_qpos, w.d_dep, drop_mask=dmf("pe_dep"), drop_scale=dsc)
        ln_bwd(w.d_dep, w.dep_pre, w.mean_d, w.rstd_d, "depth_layernorm.weight", "depth_layernorm.bias", w.d_dep_pre,
               relu=True)
        ops.gemm(GEMM_TN, w.d_dep_pre, st["x_dep"], a.g("depth_projection.weight"), ws=ws)
        ops.rowmod_sum(w.d_dep_pre, 1, a.g("depth_projection.bias").view(1, H))

    # ------------------------------------------------------------------------------------------------------
    def set_lr(self, lr):
        if self._lr_host != float(lr):
            self.lr_t.fill_(float(lr))
            self._lr_host = float(lr)

    def adamw(self, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, tick_dropout=False, ticked=False,
              skip_depth=False, prefill_dropout=False, before_flat=None):
        """One fused launch over the live prefix of the arena (main_darai.py:135; train_proposed_depth.py:215)."""
        a = self.arena
        self.set_lr(lr)
        if not ticked:
            ops.tick(self.step_t, self.drop_offset if tick_dropout else None)
        n = a.n_live
        ops.adamw_flat(a.params[:n], a.grads[:n], a.exp_avg[:n], a.exp_avg_sq[:n], self.lr_t, self.step_t, beta1=betas[0],
                       beta2=betas[1], eps=eps, weight_decay=weight_decay, grad_scale=grad_scale)

    def train_step(self, feats, depth, past_label, target_dur, target, lr, weight_decay, training=True):
        """forward + losses + backward + AdamW, all enqueued, no host sync.  Returns (loss[4], counts[4]) on device."""
        self.forward(feats, depth, past_label, "train", training)
        loss, counts = self.losses(past_label, target, target_dur, tick=True)
        self.backward()
        self.adamw(lr, weight_decay, ticked=True)
        return loss, counts
