"""Drop-in for the reference's ``model/futr_proposed.py`` (the entry-point name BASELINE.json's north star keeps): same class
name, constructor and ``forward(inputs, query, mode='train', epoch=0, idx=0)`` signature, same ``state_dict`` keys / shapes and
construction order (same seed -> same initial weights).

What it computes (futr_proposed.py:70-139): memory = relu(input_embed(x)) (no positional encoding), per-clip decoder query =
``query_embed(query)`` (nn.Embedding over label indices, query_num rows) + a sinusoidal table, the DETR decoder with the encoder
bypassed (transformer.py:77-78), adaptive average pooling of the S outputs to n_query rows, heads (``fc_seg`` has n_class - 1
outputs, :38).  The arithmetic runs in libr3d_hip.so through r3d_amd.engine_unsup.UnsupDepthEngine (label-query mode); modules
here are parameter holders.  Its training loop in the reference is train/train_unsupervised.py (out of scope, SURVEY.md
section 2): this module is driven through autograd (any torch loss on the outputs + ``.backward()``)."""
import math
import weakref

import torch
from torch import nn

from ..engine_unsup import UnsupDepthEngine
from .futr_safuser_tokenfusion import _Transformer, _PositionalEncoding


class FUTR(nn.Module):
    """FUTR(n_class, hidden_dim, src_pad_idx, device, args, n_query=8, n_head=8, num_encoder_layers=6,
    num_decoder_layers=6, query_num=48) -- model/futr_proposed.py:20-60."""

    def __init__(self, n_class, hidden_dim, src_pad_idx, device, args, n_query=8, n_head=8, num_encoder_layers=6,
                 num_decoder_layers=6, query_num=48):
        super().__init__()
        self.query_mask = query_num - 1
        self.src_pad_idx = src_pad_idx
        self.device = device
        self.hidden_dim = hidden_dim
        self.n_class = n_class
        self.n_head = n_head
        self.num_decoder_layers = num_decoder_layers
        self.n_query = n_query
        self.args = args
        if getattr(args, "input_type", "i3d_transcript") != "i3d_transcript":
            raise NotImplementedError("only input_type='i3d_transcript' is built (futr_proposed.py:57-59,93-95: 'gt' embedding)")
        if not (getattr(args, "seg", True) and getattr(args, "anticipate", True)):
            raise NotImplementedError("seg=True and anticipate=True (opts.py:100-101 defaults) are implemented")
        self.input_embed = nn.Linear(args.input_dim, hidden_dim)                                            # :27
        self.transformer = _Transformer(hidden_dim, n_head, num_encoder_layers, num_decoder_layers, hidden_dim * 4)
        nn.init.xavier_uniform_(self.input_embed.weight)
        self.query_embed = nn.Embedding(query_num, hidden_dim)                                              # :33
        self.fc_seg = nn.Linear(hidden_dim, n_class - 1)                                                    # :38
        nn.init.xavier_uniform_(self.fc_seg.weight)
        self.fc = nn.Linear(hidden_dim, n_class)
        nn.init.xavier_uniform_(self.fc.weight)
        self.fc_len = nn.Linear(hidden_dim, 1)
        nn.init.xavier_uniform_(self.fc_len.weight)
        self.pos_embedding = nn.Parameter(torch.zeros(1, args.max_pos_len, hidden_dim))
        nn.init.xavier_uniform_(self.pos_embedding)
        self.pos_enc = _PositionalEncoding(hidden_dim)                                                      # :53 (unused in forward)
        # plain tensor attribute, not a buffer (:54-55): not in the state_dict
        position = torch.arange(args.max_pos_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, hidden_dim, 2) * -(math.log(10000.0) / hidden_dim))
        pe = torch.zeros(args.max_pos_len, hidden_dim)
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.positional_embedding_l3 = pe
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self):
        dev = self.input_embed.weight.device
        if dev.type != "cuda":
            raise RuntimeError("r3d_amd.FUTR computes only on an MI355X through libr3d_hip.so; move the model to the "
                               "GPU with .to('cuda') (there is deliberately no CPU path).")
        if self._engine is None or self._engine.device != dev:
            self._engine = UnsupDepthEngine(self, dev)
            ref = weakref.ref(self._engine)
            for p in self.parameters():
                p._r3d_engine = ref
        return self._engine

    def forward(self, inputs, query, mode="train", epoch=0, idx=0):
        if mode == "train":
            src, src_label = inputs
        else:
            src, src_label = (inputs[0], None) if isinstance(inputs, (tuple, list)) else (inputs, None)
        eng = self.engine()
        src = src.to(device=eng.device, dtype=torch.float32)
        query = query.to(device=eng.device).long().contiguous()                                             # :73
        if mode == "train":
            src_label = src_label.to(device=eng.device).long().contiguous()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if not need_grad:
            out = eng.forward(src, query, src_label, mode, training=False, need_grad=False)
            return {k: v.clone() for k, v in out.items()}
        from .futr_unsupervised_depth import _Forward
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        dur, act, seg = _Forward.apply(eng, src, query, src_label, mode, self.training, names, *params)
        return {"duration": dur, "action": act, "seg": seg}
