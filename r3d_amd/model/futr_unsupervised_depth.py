"""Drop-in for the reference's ``model/futr_unsupervised_depth.py`` (the second model file BASELINE.json's north star
names): same class name, constructor and ``forward(inputs, depth_features, mode='train', epoch=0, idx=0)`` signature, same
``state_dict`` keys and shapes (incl. the two PositionalEncoding buffers and the parameters the reference never uses) and
the same construction order, so the same torch seed yields the same initial weights and reference ``.ckpt`` files load.

What it computes (futr_unsupervised_depth.py:85-163): memory = dropout(relu(input_embed(x)) + sinusoid), query =
dropout(relu(LN(depth_projection(depth 160*120))) + sinusoid) -- S queries per clip --, the DETR decoder on them with
the encoder bypassed (transformer.py:77-78), adaptive average pooling of the S outputs to n_query rows, heads.  All
arithmetic runs in libr3d_hip.so through r3d_amd.engine_unsup.UnsupDepthEngine; modules here are parameter holders.

Reference quirk, documented not emulated: in any mode but 'train' the reference's forward takes ``inputs`` as the bare
feature tensor (:91) while its own validate() (train_proposed_depth.py:72) passes the (features, labels) tuple, so the
reference crashes there.  This module accepts either.
"""
import weakref

import torch
from torch import nn

from ..engine_unsup import UnsupDepthEngine
from .futr_safuser_tokenfusion import _Transformer, _PositionalEncoding


class FUTR(nn.Module):
    """FUTR(n_class, hidden_dim, src_pad_idx, device, args, n_query=8, n_head=8, num_encoder_layers=6,
    num_decoder_layers=6, query_num=49) -- model/futr_unsupervised_depth.py:20-66."""

    def __init__(self, n_class, hidden_dim, src_pad_idx, device, args, n_query=8, n_head=8, num_encoder_layers=6,
                 num_decoder_layers=6, query_num=49, depth_pixels=160 * 120):
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
                                      "futr_unsupervised_depth.py:63-65,95-96 is not on the RGB+Depth path)")
        if not (getattr(args, "seg", True) and getattr(args, "anticipate", True)):
            raise NotImplementedError("the fused step implements seg=True and anticipate=True (opts.py:100-101 defaults)")
        if num_decoder_layers < 1:
            raise ValueError("num_decoder_layers must be >= 1")
        self.input_embed = nn.Linear(args.input_dim, hidden_dim)                                            # :28
        self.transformer = _Transformer(hidden_dim, n_head, num_encoder_layers, num_decoder_layers, hidden_dim * 4)
        nn.init.xavier_uniform_(self.input_embed.weight)
        self.l3_attention = nn.MultiheadAttention(hidden_dim, n_head, batch_first=True)                     # :34-35 (unused)
        self.query_attention = nn.MultiheadAttention(hidden_dim, n_head, batch_first=True)
        self.fc_seg = nn.Linear(hidden_dim, n_class)
        nn.init.xavier_uniform_(self.fc_seg.weight)
        self.fc = nn.Linear(hidden_dim, n_class)
        nn.init.xavier_uniform_(self.fc.weight)
        self.fc_len = nn.Linear(hidden_dim, 1)
        nn.init.xavier_uniform_(self.fc_len.weight)
        self.fc_l3 = nn.Linear(hidden_dim, query_num)                                                       # :48 (unused)
        self.pos_embedding = nn.Parameter(torch.zeros(1, args.max_pos_len, hidden_dim))
        nn.init.xavier_uniform_(self.pos_embedding)
        self.pos_enc = _PositionalEncoding(hidden_dim)                                                      # :54-55
        self.pos_enc_depth = _PositionalEncoding(hidden_dim)
        self.depth_projection = nn.Linear(depth_pixels, hidden_dim)                                         # :59
        nn.init.xavier_uniform_(self.depth_projection.weight)
        self.depth_layernorm = nn.LayerNorm(hidden_dim)
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self):
        dev = self.depth_projection.weight.device
        if dev.type != "cuda":
            raise RuntimeError("r3d_amd.FUTR computes only on an MI355X through libr3d_hip.so; move the model to the "
                               "GPU with .to('cuda') (there is deliberately no CPU path).")
        if self._engine is None or self._engine.device != dev:
            self._engine = UnsupDepthEngine(self, dev)
            ref = weakref.ref(self._engine)
            for p in self.parameters():
                p._r3d_engine = ref
        return self._engine

    def forward(self, inputs, depth_features, mode="train", epoch=0, idx=0):
        if mode == "train":
            src, src_label = inputs
        else:
            src, src_label = (inputs[0], None) if isinstance(inputs, (tuple, list)) else (inputs, None)
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
        dur, act, seg = _Forward.apply(eng, src, depth_features, src_label, mode, self.training, names, *params)
        return {"duration": dur, "action": act, "seg": seg}


class _Forward(torch.autograd.Function):
    """Bridges the engine into autograd (losses on the outputs, losses.backward(), any torch optimiser)."""

    @staticmethod
    def forward(ctx, eng, src, depth, labels, mode, training, names, *params):
        out = eng.forward(src, depth, labels, mode, training=training, need_grad=True)
        ctx.eng, ctx.names, ctx.token = eng, names, eng.last
        return out["duration"].clone(), out["action"].clone(), out["seg"].clone()

    @staticmethod
    def backward(ctx, d_dur, d_act, d_seg):
        eng = ctx.eng
        if eng.last is not ctx.token:
            raise RuntimeError("r3d_amd: backward() must follow the forward() it belongs to")
        w, K = eng.last["w"], eng.K
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
            w.d_seg.copy_(d_seg.reshape(w.d_seg.shape))
        eng.backward()
        grads = [eng.arena.g(n).clone() if eng.arena.is_live(n) else None for n in ctx.names]
        return (None,) * 7 + tuple(grads)
