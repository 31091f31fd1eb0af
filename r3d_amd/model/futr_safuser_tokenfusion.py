"""Drop-in for the reference's ``model/futr_safuser_tokenfusion.py``: same class names (CMFuser, FUTR), constructor and
``forward(inputs, depth_features, mode='train', epoch=0, idx=0)`` signature, same ``state_dict`` keys and shapes
(including the parameters the reference never uses) and the same construction order, so the same
torch seed yields the same initial weights and reference ``.ckpt`` files load.

The nn.Linear / nn.LayerNorm / nn.MultiheadAttention / nn.Embedding objects below are PARAMETER HOLDERS only: their
forward is never called.  All arithmetic runs in the HIP kernels of libr3d_hip.so through r3d_amd.engine.FusionEngine;
on a non-HIP device forward() raises (there is no CPU path).
"""
import copy
import math
import weakref

import torch
from torch import nn

from ..engine import FusionEngine


# ---- parameter holders with the reference's module tree (model/extras/transformerblock.py, transformer.py) ----------
class _Attention(nn.Module):
    def __init__(self, dim, qkv_bias=False):                      # transformerblock.py:8-17
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class _MLP(nn.Module):
    def __init__(self, dim, hidden):                               # transformerblock.py:80-90 (keys mlp.0 / mlp.2)
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim), nn.Dropout(0.0))


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False):   # transformerblock.py:119-129
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attention(dim, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _MLP(dim, int(dim * mlp_ratio))


class _EncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_ff, dropout=0.1):      # transformer.py:195-211 (constructed, never run)
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_ff)
        self.linear2 = nn.Linear(dim_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)


class _DecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_ff, dropout=0.1):      # transformer.py:256-276
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_ff)
        self.linear2 = nn.Linear(dim_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)


class _Stack(nn.Module):
    def __init__(self, layer, n, norm=None):                       # transformer.py:130-137,152-159 (_get_clones)
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])
        self.norm = norm


class _Transformer(nn.Module):
    def __init__(self, d_model, nhead, n_enc, n_dec, dim_ff, dropout=0.1):    # transformer.py:22-44
        super().__init__()
        self.encoder = _Stack(_EncoderLayer(d_model, nhead, dim_ff, dropout), n_enc, None)
        self.decoder = _Stack(_DecoderLayer(d_model, nhead, dim_ff, dropout), n_dec, nn.LayerNorm(d_model))
        for p in self.parameters():                                # _reset_parameters (:70-73)
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class _PositionalEncoding(nn.Module):
    def __init__(self, d_model, max_len=3000):                     # position.py:17-27 (buffer only; unused on the path)
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pos_table", pe)


class CMFuser(nn.Module):
    """SA-Fuser parameter tree (model/futr_safuser_tokenfusion.py:17-31).  Token selection, exchange, the fuser block
    and the modality mean run inside FusionEngine.forward; this class only owns the parameters."""

    def __init__(self, dim, depth=1, num_heads=4, mlp_ratio=4.0, qkv_bias=False):
        super().__init__()
        if depth != 1:
            raise NotImplementedError("the reference builds CMFuser(depth=1) (futr_safuser_tokenfusion.py:120)")
        self.blocks = nn.ModuleList([_Block(dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)
        self.modality_token = nn.Parameter(torch.randn(1, 1, 1, dim))
        self.projection = nn.Linear(dim, dim)
        self.fusion_conv = nn.Conv2d(in_channels=2, out_channels=1, kernel_size=1)


class FUTR(nn.Module):
    """FUTR(n_class, hidden_dim, src_pad_idx, device, args, n_query=8, n_head=8, num_encoder_layers=6,
    num_decoder_layers=6, query_num=49) -- model/futr_safuser_tokenfusion.py:103-152."""

    _fuser_cls = CMFuser                # (the BN-blend variant swaps in its own fuser, futr_safuser_batchnormalization.py)

    def __init__(self, n_class, hidden_dim, src_pad_idx, device, args, n_query=8, n_head=8, num_encoder_layers=6,
                 num_decoder_layers=6, query_num=49, depth_pixels=224 * 224):
        super().__init__()
        self.src_pad_idx = src_pad_idx
        self.query_pad_idx = query_num - 1
        self.device = device
        self.hidden_dim = hidden_dim
        self.n_class = n_class
        self.n_head = n_head
        self.num_decoder_layers = num_decoder_layers
        self.n_query = n_query
        self.args = args
        if getattr(args, "input_type", "i3d_transcript") != "i3d_transcript":
            raise NotImplementedError("only input_type='i3d_transcript' is built (the 'gt' embedding branch of "
                                      "futr_safuser_tokenfusion.py:150-152,180-182 is not on the RGB+Depth path)")
        if not (getattr(args, "seg", True) and getattr(args, "anticipate", True)):
            raise NotImplementedError("the fused step implements seg=True and anticipate=True (opts.py:100-101 defaults)")
        if num_decoder_layers < 1:
            raise ValueError("num_decoder_layers must be >= 1")
        self.input_embed = nn.Linear(args.input_dim, hidden_dim)
        self.transformer = _Transformer(hidden_dim, n_head, num_encoder_layers, num_decoder_layers, hidden_dim * 4)
        nn.init.xavier_uniform_(self.input_embed.weight)
        self.l3_attention = nn.MultiheadAttention(hidden_dim, n_head, batch_first=True)
        self.query_attention = nn.MultiheadAttention(hidden_dim, n_head, batch_first=True)
        self.query_embed = nn.Embedding(self.n_query, hidden_dim)
        self.fuser = self._fuser_cls(dim=hidden_dim, depth=1, num_heads=n_head)
        self.fc_seg = nn.Linear(hidden_dim, n_class)
        nn.init.xavier_uniform_(self.fc_seg.weight)
        self.fc = nn.Linear(hidden_dim, n_class)
        nn.init.xavier_uniform_(self.fc.weight)
        self.fc_len = nn.Linear(hidden_dim, 1)
        nn.init.xavier_uniform_(self.fc_len.weight)
        self.fc_l3 = nn.Linear(hidden_dim, query_num)
        self.pos_embedding = nn.Parameter(torch.zeros(1, args.max_pos_len, hidden_dim))
        nn.init.xavier_uniform_(self.pos_embedding)
        self.pos_enc = _PositionalEncoding(hidden_dim)
        self.pos_enc_depth = _PositionalEncoding(hidden_dim)
        # 224*224 for DARai / NTU (:143); 160*120 for UTKinect is the commented alternative (:144)
        self.depth_projection = nn.Linear(depth_pixels, hidden_dim)
        nn.init.xavier_uniform_(self.depth_projection.weight)
        self.depth_layernorm = nn.LayerNorm(hidden_dim)
        self._engine = None

    # ---- engine life cycle ------------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._engine = None                     # .to()/.cuda() re-creates parameter storage; re-flatten lazily
        return super()._apply(fn, *a, **k)

    def engine(self):
        dev = self.depth_projection.weight.device
        if dev.type != "cuda":
            raise RuntimeError("r3d_amd.FUTR computes only on an MI355X through libr3d_hip.so; move the model to the "
                               "GPU with .to('cuda') (there is deliberately no CPU path).")
        if self._engine is None or self._engine.device != dev:
            self._engine = FusionEngine(self, dev)
            ref = weakref.ref(self._engine)
            for p in self.parameters():
                p._r3d_engine = ref              # lets r3d_amd.optim.FlatAdamW find the arena behind its parameters
        return self._engine

    # ---- forward ------------------------------------------------------------------------------------------------
    def forward(self, inputs, depth_features, mode="train", epoch=0, idx=0):
        src, src_label = inputs
        eng = self.engine()
        src = src.to(device=eng.device, dtype=torch.float32)
        depth_features = depth_features.to(device=eng.device, dtype=torch.float32)
        if mode == "train":
            src_label = src_label.to(device=eng.device).long().contiguous()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if not need_grad:
            out = eng.forward(src, depth_features, src_label, mode, training=False, need_grad=False)
            return {k: v.clone() for k, v in out.items()}
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        dur, act, seg = _FusedForward.apply(eng, src, depth_features, src_label, mode, self.training, names, *params)
        return {"duration": dur, "action": act, "seg": seg}


class _FusedForward(torch.autograd.Function):
    """Bridges the engine into autograd so that the reference's own loop (loss on the outputs, losses.backward(),
    any torch optimiser) works unchanged on this module."""

    @staticmethod
    def forward(ctx, eng, src, depth, labels, mode, training, names, *params):
        keep, eng.defer_tail = eng.defer_tail, False        # the caller reads the outputs before any loss exists
        try:
            out = eng.forward(src, depth, labels, mode, training=training, need_grad=True)
        finally:
            eng.defer_tail = keep
        ctx.eng, ctx.names, ctx.token = eng, names, eng.last
        return out["duration"].clone(), out["action"].clone(), out["seg"].clone()

    @staticmethod
    def backward(ctx, d_dur, d_act, d_seg):
        eng = ctx.eng
        if eng.last is not ctx.token:
            raise RuntimeError("r3d_amd: backward() must follow the forward() it belongs to (the engine keeps one "
                               "set of saved activations per shape)")
        w = eng.last["w"]
        K = eng.K
        if d_act is None:
            w.d_actdur[:, :K].zero_()
        else:
            w.d_actdur[:, :K].copy_(d_act.reshape(-1, K))
        if d_dur is None:
            w.d_actdur[:, K].zero_()
        else:
            w.d_actdur[:, K].copy_(d_dur.reshape(-1))
        if d_seg is None:
            w.d_seg.zero_()
        else:
            w.d_seg.copy_(d_seg.reshape(-1, K))
        eng.backward()
        grads = [eng.arena.g(n).clone() if eng.arena.is_live(n) else None for n in ctx.names]
        return (None,) * 7 + tuple(grads)

