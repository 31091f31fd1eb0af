"""TEST INFRASTRUCTURE -- CPU restatement (PyTorch CPU ops) of R3D's RGB+Depth token-fusion
training step.  It is the checker for the HIP path and the timed ``cpu_baseline`` ("port").
It is NOT imported by the product package.

Each function cites the reference lines it restates (paths relative to /root/reference).
Parameters are a plain ``dict name -> tensor`` using the reference's ``state_dict`` names
(SURVEY.md Appendix B), so reference checkpoints and the golden fixtures map 1:1.

Pinned by tests/golden/*.npz (made by tests/golden/make_golden.py from the imported reference).
Dropout is not restated (no RNG parity is possible); the oracle is the reference in
``model.eval()`` state, which is also the reference's de-facto training state after its first
``validate()`` call (train/train_proposed_depth.py:53 is never undone).
"""
import math
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5


# ----------------------------------------------------------------------------------------
# token selection: torch.topk(score, k, largest=False) on CPU
#   (model/futr_safuser_tokenfusion.py:52-54).  ATen's CPU top-k for k*64 > n uses
#   std::nth_element(queue, queue+k-1, end, value-only '<') over (value, index) pairs
#   (aten/src/ATen/native/cpu/TopKImpl.h, topk_impl_loop) -- libstdc++'s introselect.
#   Ties are therefore resolved by introselect's data movement; it is restated here so the
#   selected index SET is bit-exact even when every score is equal (train mode).
# ----------------------------------------------------------------------------------------
def _introselect_py(vals, k):
    """Pure-python restatement of libstdc++ std::nth_element (bits/stl_algo.h: __introselect,
    __unguarded_partition_pivot, __move_median_to_first, __heap_select, __insertion_sort) with
    comparator (a.value < b.value) [NaN sorts last], on (value, index) pairs.
    Returns the first k indices after the call (the selected set, unsorted)."""
    n = len(vals)
    v = [float(x) for x in vals]
    idx = list(range(n))

    def less(a, b):  # positions a, b
        x, y = v[a], v[b]
        return ((not math.isnan(x)) and math.isnan(y)) or (x < y)

    def swap(a, b):
        v[a], v[b] = v[b], v[a]
        idx[a], idx[b] = idx[b], idx[a]

    def less_val(xv, b):  # value xv < element at b
        y = v[b]
        return ((not math.isnan(xv)) and math.isnan(y)) or (xv < y)

    def val_less(a, yv):
        x = v[a]
        return ((not math.isnan(x)) and math.isnan(yv)) or (x < yv)

    def adjust_heap(first, hole, length, val, vidx):
        top = hole
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if less(first + child, first + child - 1):
                child -= 1
            v[first + hole], idx[first + hole] = v[first + child], idx[first + child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            v[first + hole], idx[first + hole] = v[first + child - 1], idx[first + child - 1]
            hole = child - 1
        # __push_heap
        parent = (hole - 1) // 2
        while hole > top and val_less(first + parent, val):
            v[first + hole], idx[first + hole] = v[first + parent], idx[first + parent]
            hole = parent
            parent = (hole - 1) // 2
        v[first + hole], idx[first + hole] = val, vidx

    def make_heap(first, last):
        length = last - first
        if length < 2:
            return
        parent = (length - 2) // 2
        while True:
            adjust_heap(first, parent, length, v[first + parent], idx[first + parent])
            if parent == 0:
                return
            parent -= 1

    def heap_select(first, middle, last):
        make_heap(first, middle)
        for i in range(middle, last):
            if less(i, first):
                val, vi = v[i], idx[i]
                v[i], idx[i] = v[first], idx[first]
                adjust_heap(first, 0, middle - first, val, vi)

    def insertion_sort(first, last):
        if first == last:
            return
        for i in range(first + 1, last):
            if less(i, first):
                val, vi = v[i], idx[i]
                for j in range(i, first, -1):
                    v[j], idx[j] = v[j - 1], idx[j - 1]
                v[first], idx[first] = val, vi
            else:
                val, vi = v[i], idx[i]
                j = i
                while less_val(val, j - 1):
                    v[j], idx[j] = v[j - 1], idx[j - 1]
                    j -= 1
                v[j], idx[j] = val, vi

    first, last, nth = 0, n, k - 1
    depth = 2 * (n.bit_length() - 1) if n > 0 else 0
    while last - first > 3:
        if depth == 0:
            heap_select(first, nth + 1, last)
            swap(first, nth)
            return idx[:k]
        depth -= 1
        mid = first + (last - first) // 2
        a, b, c = first + 1, mid, last - 1
        if less(a, b):
            if less(b, c):
                swap(first, b)
            elif less(a, c):
                swap(first, c)
            else:
                swap(first, a)
        elif less(a, c):
            swap(first, a)
        elif less(b, c):
            swap(first, c)
        else:
            swap(first, b)
        lo, hi, piv = first + 1, last, first
        while True:
            while less(lo, piv):
                lo += 1
            hi -= 1
            while less(piv, hi):
                hi -= 1
            if not (lo < hi):
                break
            swap(lo, hi)
            lo += 1
        cut = lo
        if cut <= nth:
            first = cut
        else:
            last = cut
    insertion_sort(first, last)
    return idx[:k]


_TOPK_LIB = None


def _topk_lib():
    """oracle/_build/libr3d_oracle.so (built from oracle/topk_introselect.c by oracle/Makefile)."""
    global _TOPK_LIB
    if _TOPK_LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libr3d_oracle.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.r3d_oracle_select_smallest.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                       ctypes.c_void_p]
            lib.r3d_oracle_select_smallest.restype = ctypes.c_int
            _TOPK_LIB = lib
        else:
            _TOPK_LIB = False
    return _TOPK_LIB


def select_smallest(score, k, use_c=True):
    """Index set (sorted ascending, int64 ndarray) that torch.topk(score, k, largest=False) returns on CPU."""
    s = np.ascontiguousarray(np.asarray(score, dtype=np.float32).reshape(-1))
    lib = _topk_lib() if use_c else False
    if lib:
        out = np.empty(k, dtype=np.int64)
        rc = lib.r3d_oracle_select_smallest(s.ctypes.data, s.size, k, out.ctypes.data)
        assert rc == 0
        return np.sort(out)
    return np.sort(np.asarray(_introselect_py(s.tolist(), k), dtype=np.int64))


# ----------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------
def layer_norm(x, w, b):
    # nn.LayerNorm defaults: eps 1e-5, biased variance
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def mha(p, prefix, query, key, value, n_head, key_padding_mask=None):
    """nn.MultiheadAttention (seq-first inputs are passed here batch-first: [B,L,H]).
    model/extras/transformer.py:275-276, 289-304; SURVEY Appendix A.5.  Returns [B,Lq,H]."""
    H = query.shape[-1]
    dh = H // n_head
    w, bias = p[prefix + "in_proj_weight"], p[prefix + "in_proj_bias"]
    q = F.linear(query, w[:H], bias[:H])
    k = F.linear(key, w[H:2 * H], bias[H:2 * H])
    v = F.linear(value, w[2 * H:], bias[2 * H:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    q = q.view(B, Lq, n_head, dh).transpose(1, 2) * (dh ** -0.5)
    k = k.view(B, Lk, n_head, dh).transpose(1, 2)
    v = v.view(B, Lk, n_head, dh).transpose(1, 2)
    att = q @ k.transpose(-2, -1)                                  # [B,h,Lq,Lk]
    if key_padding_mask is not None:
        att = att.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, Lq, H)
    return F.linear(o, p[prefix + "out_proj.weight"], p[prefix + "out_proj.bias"])


def token_fusion(rgb, dep, mode, world_scores=None):
    """CMFuser.token_fusion (model/futr_safuser_tokenfusion.py:33-66).
    train: score = |d(mean(rgb)+mean(dep))/dx| averaged over (B,T) == the constant 1/(B*T*C)
           for every channel (:40-45) -> top-k is pure tie-breaking;
    else : score = mean_(B,T) |x| per channel (:47-50).  k = C//4 smallest (:52-54); hard channel
    swap on clones (:56-60); stack to [B,T,2,C] (:62)."""
    B, T, C = rgb.shape
    if mode == "train":
        const = torch.full((C,), 1.0 / (B * T * C), dtype=torch.float32)
        s_rgb, s_dep = const, const
    else:
        s_rgb = rgb.detach().abs().mean(dim=(0, 1))
        s_dep = dep.detach().abs().mean(dim=(0, 1))
    k = C // 4
    idx_rgb = torch.from_numpy(select_smallest(s_rgb.numpy(), k))
    idx_dep = torch.from_numpy(select_smallest(s_dep.numpy(), k))
    m_rgb = torch.zeros(C, dtype=torch.bool)
    m_dep = torch.zeros(C, dtype=torch.bool)
    m_rgb[idx_rgb] = True
    m_dep[idx_dep] = True
    ex_rgb = torch.where(m_rgb, dep, rgb)
    ex_dep = torch.where(m_dep, rgb, dep)
    return torch.stack([ex_rgb, ex_dep], dim=2), idx_rgb, idx_dep, s_rgb, s_dep


def token_fusion_bn(p, state, rgb, dep, training, momentum=0.1, eps=1e-5):
    """CMFuser.token_fusion of the BN-blend variant (model/futr_safuser_batchnormalization.py:38-76):
    BatchNorm1d over (B,T) per channel on both embeddings (:45-46; batch statistics + running-stat update when
    training, running statistics otherwise), score = |BN gamma| (:48-49), k = int(0.1 C) smallest (:58-60),
    exchanged = alpha * own + (1 - alpha) * other on the selected channels (:65-74), stack (:75).
    state: dict of the BatchNorm buffers (running_mean / running_var / num_batches_tracked), updated in place."""
    B, T, C = rgb.shape

    def bn(x, pre):
        y = F.batch_norm(x.permute(0, 2, 1), state[pre + "running_mean"], state[pre + "running_var"], p[pre + "weight"],
                         p[pre + "bias"], training, momentum, eps)
        if training:
            state[pre + "num_batches_tracked"] += 1
        return y.permute(0, 2, 1)
    r, d = bn(rgb, "fuser.bn_rgb."), bn(dep, "fuser.bn_depth.")
    k = max(0, int(C * 0.1))
    g_r, g_d = p["fuser.bn_rgb.weight"].detach().abs(), p["fuser.bn_depth.weight"].detach().abs()
    idx_rgb = torch.from_numpy(select_smallest(g_r.numpy(), k))
    idx_dep = torch.from_numpy(select_smallest(g_d.numpy(), k))
    m_rgb = torch.zeros(C, dtype=torch.bool)
    m_dep = torch.zeros(C, dtype=torch.bool)
    m_rgb[idx_rgb] = True
    m_dep[idx_dep] = True
    alpha = p["fuser.alpha"].view(C)
    ex_rgb = torch.where(m_rgb, alpha * r + (1 - alpha) * d, r)
    ex_dep = torch.where(m_dep, alpha * d + (1 - alpha) * r, d)
    return torch.stack([ex_rgb, ex_dep], dim=2), idx_rgb, idx_dep, g_r, g_d


def cm_fuser_bn(p, state, rgb, dep, training, n_head):
    """CMFuser.forward of the BN-blend variant (futr_safuser_batchnormalization.py:84-107): no x_res (:97,101)."""
    B, T, C = rgb.shape
    stacked, idx_rgb, idx_dep, s_rgb, s_dep = token_fusion_bn(p, state, rgb, dep, training)
    x = fuser_block(p, stacked.reshape(B * T, 2, C), n_head)
    x = layer_norm(x, p["fuser.norm.weight"], p["fuser.norm.bias"])
    fused = x.mean(dim=1).view(B, T, C)
    return fused, dict(idx_rgb=idx_rgb, idx_dep=idx_dep, score_rgb=s_rgb, score_dep=s_dep)


def fuser_block(p, x, n_head):
    """Block (model/extras/transformerblock.py:118-135) with Attention (:7-36) under the additive
    mask [[-inf,0],[0,-inf]] (futr_safuser_tokenfusion.py:68-72,77) and MLP (:79-93, exact-erf GELU).
    x: [N,2,C].  The full q/k/softmax path is kept here on purpose (the product uses the closed form)."""
    pre = "fuser.blocks.0."
    Nn, M, C = x.shape
    dh = C // n_head
    h1 = layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
    qkv = F.linear(h1, p[pre + "attn.qkv.weight"]).reshape(Nn, M, 3, n_head, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    mask = torch.eye(M)
    mask = mask.masked_fill(mask == 1, float("-inf"))
    att = (att + mask).softmax(dim=-1)
    a = (att @ v).transpose(1, 2).reshape(Nn, M, C)
    a = F.linear(a, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
    x = x + a
    h2 = layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
    f = F.linear(h2, p[pre + "mlp.mlp.0.weight"], p[pre + "mlp.mlp.0.bias"])
    f = F.gelu(f)
    f = F.linear(f, p[pre + "mlp.mlp.2.weight"], p[pre + "mlp.mlp.2.bias"])
    return x + f


def cm_fuser(p, rgb, dep, mode, n_head):
    """CMFuser.forward (futr_safuser_tokenfusion.py:74-97), dropout omitted (eval state)."""
    B, T, C = rgb.shape
    stacked, idx_rgb, idx_dep, s_rgb, s_dep = token_fusion(rgb, dep, mode)
    x = stacked.reshape(B * T, 2, C)
    x_res = x
    x = fuser_block(p, x, n_head)
    x = x + x_res
    x = layer_norm(x, p["fuser.norm.weight"], p["fuser.norm.bias"])
    fused = x.mean(dim=1).view(B, T, C)
    return fused, dict(idx_rgb=idx_rgb, idx_dep=idx_dep, score_rgb=s_rgb, score_dep=s_dep)


def cm_fuser_m(p, xs, mode, n_head):
    """BUILD-DEFINED M-modality extension of CMFuser.forward (BASELINE.json configs[4]) -- **parity
    unpinned**: the reference's CMFuser is two-token (futr_safuser_tokenfusion.py:74-81 hard-codes 'rgb' / 'depth' and a
    2 x 2 mask), so no reference output exists for M = 3.  Every line that generalises is kept: the per-modality scores and
    k = C // 4 of token_fusion (:40-54), the exchange with the NEXT modality cyclically (m takes the selected channels of
    (m + 1) mod M -- for M = 2 exactly :56-60), generate_cross_attention_mask(M) (-inf diagonal, :68-72; fuser_block builds
    it for any M), Block, + x_res, norm, mean over the M tokens (:83-94).  For M = 2 this function IS cm_fuser
    (tests/test_oracle_golden.py checks it bit for bit), which the reference fixtures pin."""
    B, T, C = xs[0].shape
    M = len(xs)
    k = C // 4
    if mode == "train":
        scores = [torch.full((C,), 1.0 / (B * T * C), dtype=torch.float32) for _ in xs]
    else:
        scores = [x.detach().abs().mean(dim=(0, 1)) for x in xs]
    idx = [torch.from_numpy(select_smallest(s_.numpy(), k)) for s_ in scores]
    ex = []
    for m in range(M):
        mk = torch.zeros(C, dtype=torch.bool)
        mk[idx[m]] = True
        ex.append(torch.where(mk, xs[(m + 1) % M], xs[m]))
    x = torch.stack(ex, dim=2).reshape(B * T, M, C)
    x_res = x
    x = fuser_block(p, x, n_head)
    x = x + x_res
    x = layer_norm(x, p["fuser.norm.weight"], p["fuser.norm.bias"])
    return x.mean(dim=1).view(B, T, C), dict(idx=idx, scores=scores)


def decoder(p, memory, pos, query_pos, key_padding_mask, n_head, n_layers, capture=None):
    """Transformer.forward with the encoder bypassed (model/extras/transformer.py:75-128) ->
    TransformerDecoder (:161-191) -> TransformerDecoderLayer.forward_post (:281-330), post-norm.
    All tensors batch-first here: memory [B,S,H], pos [1,S,H], query_pos [1,Q,H]."""
    B = memory.shape[0]
    qp = query_pos.expand(B, -1, -1)
    tgt = torch.zeros_like(qp)
    kv = memory + pos                                                 # key AND value carry pos (:300-302)
    for l in range(n_layers):
        pre = f"transformer.decoder.layers.{l}."
        qk = tgt + qp
        sa = mha(p, pre + "self_attn.", qk, qk, qk, n_head)           # v also carries query_pos (:289)
        tgt = layer_norm(tgt + sa, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
        ca = mha(p, pre + "multihead_attn.", tgt + qp, kv, kv, n_head, key_padding_mask)
        tgt = layer_norm(tgt + ca, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
        ff_pre = F.linear(tgt, p[pre + "linear1.weight"], p[pre + "linear1.bias"])
        if capture is not None:                                       # (tests: which ReLU units sit on the kink)
            capture.setdefault("ffn_pre", []).append(ff_pre.detach())
        ff = F.linear(F.relu(ff_pre), p[pre + "linear2.weight"], p[pre + "linear2.bias"])
        tgt = layer_norm(tgt + ff, p[pre + "norm3.weight"], p[pre + "norm3.bias"])
    return layer_norm(tgt, p["transformer.decoder.norm.weight"], p["transformer.decoder.norm.bias"])


def forward(p, inputs, depth, mode, pad_idx, n_head=8, n_layers=1, want_seg=True, want_anticipate=True, bn_state=None,
            bn_training=False):
    """FUTR.forward (model/futr_safuser_tokenfusion.py:164-239), input_type 'i3d_transcript'.
    bn_state (dict of BatchNorm buffers) selects the BN-blend fuser of model/futr_safuser_batchnormalization.py (same
    FUTR around it, :110-270); bn_training = module.training.
    Returns (outputs dict, aux dict with fused features / selected indices)."""
    src, src_label = inputs
    B, S, _ = src.shape
    kpm = (src_label == pad_idx) if mode == "train" else None                       # :165-174
    rgb = F.relu(F.linear(src, p["input_embed.weight"], p["input_embed.bias"]))      # :179-183
    pos = p["pos_embedding"][:, :S]                                                  # :190
    d = depth.reshape(B, S, -1)                                                      # :194
    d = F.linear(d, p["depth_projection.weight"], p["depth_projection.bias"])        # :195
    d = F.relu(layer_norm(d, p["depth_layernorm.weight"], p["depth_layernorm.bias"]))  # :196-197
    if bn_state is not None:
        fused, aux = cm_fuser_bn(p, bn_state, rgb, d, bn_training, n_head)
    else:
        fused, aux = cm_fuser(p, rgb, d, mode, n_head)                               # :199
    qpos = p["query_embed.weight"].unsqueeze(0)                                      # :205-209
    tgt = decoder(p, fused, pos, qpos, kpm, n_head, n_layers, capture=aux)          # :211
    out = {}
    if want_anticipate:                                                              # :219-226
        out["action"] = F.linear(tgt, p["fc.weight"], p["fc.bias"])
        out["duration"] = F.linear(tgt, p["fc_len.weight"], p["fc_len.bias"]).squeeze(2)
    if want_seg:                                                                     # :228-232
        out["seg"] = F.linear(fused, p["fc_seg.weight"], p["fc_seg.bias"])
    aux["fused"] = fused
    aux["rgb_embed"] = rgb
    aux["depth_embed"] = d
    return out, aux


# ----------------------------------------------------------------------------------------
# depth-as-query model (model/futr_unsupervised_depth.py)
# ----------------------------------------------------------------------------------------
def sinusoid_table(max_len, d_model):
    """PositionalEncoding's buffer `pos_table` [1, max_len, d_model] (model/extras/position.py:19-27)."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def forward_unsup_depth(p, inputs, depth, mode, pad_idx, n_head=8, n_layers=1, n_query=8):
    """FUTR.forward of model/futr_unsupervised_depth.py:85-163, input_type 'i3d_transcript', dropout omitted
    (eval state).  The RGB embedding + sinusoidal encoding is the decoder's MEMORY (the encoder is bypassed,
    transformer.py:77-78); the depth embedding + sinusoidal encoding is its per-clip QUERY (S queries per clip, :115,
    :128); the S decoder outputs are average-pooled to n_query rows (:134) before the anticipation heads.
    In any mode but 'train' the reference takes `inputs` as the bare feature tensor (:90) -- its own validate() passes a
    tuple there and crashes; a tuple is accepted here."""
    if mode == "train":
        src, src_label = inputs                                                      # :87
        kpm = (src_label == pad_idx)                                                 # :89
    else:
        src = inputs[0] if isinstance(inputs, (tuple, list)) else inputs            # :91
        kpm = None
    B, S, _ = src.shape
    H = p["input_embed.weight"].shape[0]
    pe = sinusoid_table(3000, H)[:, :S]
    mem = F.relu(F.linear(src, p["input_embed.weight"], p["input_embed.bias"])) + pe   # :96-99 (pos_enc, dropout omitted)
    pos = p["pos_embedding"][:, :S]                                                  # :104
    d = depth.reshape(B, S, -1)                                                      # :107-108 (depth is [B,S,Hh,Ww] here)
    d = F.linear(d, p["depth_projection.weight"], p["depth_projection.bias"])        # :109
    d = F.relu(layer_norm(d, p["depth_layernorm.weight"], p["depth_layernorm.bias"]))  # :110-111
    query = d + pe                                                                   # :115 (pos_enc_depth)
    tgt = decoder(p, mem, pos, query, kpm, n_head, n_layers)                         # :128, tgt = zeros_like(query) :126
    pooled = F.adaptive_avg_pool1d(tgt.permute(0, 2, 1), n_query).permute(0, 2, 1)   # :134
    out = {"action": F.linear(pooled, p["fc.weight"], p["fc.bias"]),                  # :140-144
           "duration": F.linear(pooled, p["fc_len.weight"], p["fc_len.bias"]).squeeze(2),
           "seg": F.linear(mem, p["fc_seg.weight"], p["fc_seg.bias"])}               # :148 (src returned unchanged = memory)
    return out, dict(memory=mem, query=query, tgt=tgt, pooled=pooled)


def forward_proposed(p, inputs, query, mode, pad_idx, n_head=8, n_layers=1, n_query=8):
    """FUTR.forward of model/futr_proposed.py:70-139 (input_type 'i3d_transcript'): memory = relu(input_embed(x)) with no
    positional encoding (:92-97), per-clip query = query_embed(label indices) + sinusoidal table (:103-106), decoder with the
    encoder bypassed (:118), adaptive average pooling to n_query rows (:124), heads; fc_seg has n_class - 1 outputs (:38)."""
    if mode == "train":
        src, src_label = inputs
        kpm = (src_label == pad_idx)
    else:
        src = inputs[0] if isinstance(inputs, (tuple, list)) else inputs
        kpm = None
    B, S, _ = src.shape
    H = p["input_embed.weight"].shape[0]
    mem = F.relu(F.linear(src, p["input_embed.weight"], p["input_embed.bias"]))
    q = F.embedding(query.long(), p["query_embed.weight"]) + sinusoid_table(S, H)     # positional_embedding_l3[:S] (:63-69,104-106)
    pos = p["pos_embedding"][:, :S]
    tgt = decoder(p, mem, pos, q, kpm, n_head, n_layers)
    pooled = F.adaptive_avg_pool1d(tgt.permute(0, 2, 1), n_query).permute(0, 2, 1)
    out = {"action": F.linear(pooled, p["fc.weight"], p["fc.bias"]),
           "duration": F.linear(pooled, p["fc_len.weight"], p["fc_len.bias"]).squeeze(2),
           "seg": F.linear(mem, p["fc_seg.weight"], p["fc_seg.bias"])}
    return out, dict(memory=mem, query=q, tgt=tgt, pooled=pooled)


UNSUP_LIVE_PREFIXES = ("input_embed.", "depth_projection.", "depth_layernorm.", "pos_embedding", "transformer.decoder.",
                       "fc_seg.", "fc.", "fc_len.")


# ----------------------------------------------------------------------------------------
# losses (utils.py:325-328, 358-378, 410-490; train/train_proposed_depth.py:28-50, 171-213)
# ----------------------------------------------------------------------------------------
EXCLUDE_CLASS_IDX = 47   # hard-coded in train_proposed_depth.py:181,195


def last_non_padding_labels(past_label, pad_value):
    """get_last_non_padding_labels (train_proposed_depth.py:28-50)."""
    B = past_label.shape[0]
    out = torch.zeros(B, dtype=past_label.dtype)
    for i in range(B):
        nz = (past_label[i] != pad_value).nonzero(as_tuple=True)[0]
        out[i] = past_label[i, nz[-1]] if nz.numel() > 0 else pad_value
    return out


def masked_ce(pred, gold, pad_idx, exclude=EXCLUDE_CLASS_IDX):
    """shared part of cal_loss / cal_weighted_loss (utils.py:425-433, 470-477): labels equal to pad or
    the excluded class -> ignore (-1); per-row CE, ignored rows contribute 0."""
    gold = gold.long()
    mask = (gold != pad_idx) & (gold != exclude)
    g = gold.clone()
    g[~mask] = -1
    return F.cross_entropy(pred, g, ignore_index=-1, reduction="none"), mask


def seg_loss(pred, gold, pad_idx):
    """cal_loss (utils.py:449-490): + 2.0 * [argmax == pad_idx & valid]; mean over ALL rows."""
    base, mask = masked_ce(pred, gold, pad_idx)
    penalty = 2.0 * ((pred.argmax(dim=1) == pad_idx) & mask).float()
    return (base + penalty).mean()


def action_loss(pred, gold, pad_idx, t_n_labels, target_ref):
    """cal_weighted_loss (utils.py:410-447): per-clip weight 1 if last observed label == first future
    label else 10, repeat_interleave over the Q queries, mean over all rows."""
    base, _ = masked_ce(pred, gold, pad_idx)
    w = torch.where(t_n_labels == target_ref, 1.0, 10.0)
    w = w.repeat_interleave(base.shape[0] // w.shape[0])
    return (base * w).mean()


def counts(pred, gold, pad_idx, exclude=EXCLUDE_CLASS_IDX):
    """n_correct / n_word of cal_performance (utils.py:368-376)."""
    gold = gold.long()
    m = gold.ne(pad_idx) & gold.ne(exclude)
    return int(pred.max(1)[1].eq(gold).masked_select(m).sum()), int(m.sum())


def normalize_duration(x, mask):
    """utils.py:325-328."""
    return F.normalize(torch.exp(x) * mask, p=1, dim=-1)


def losses(out, past_label, trans_dur_future, trans_future_target, pad_idx, dur_den=None):
    """The loss composition of train_proposed_depth.py:139-213.  Returns dict of scalars + counts.
    dur_den overrides the duration-loss denominator (data-parallel shards use global mask sum / world size)."""
    B = trans_dur_future.shape[0]
    dur_mask = (trans_dur_future != pad_idx).long()
    target_dur = trans_dur_future * dur_mask
    res = {}
    total = 0.0
    seg = out["seg"]
    K = seg.shape[-1]
    l_seg = seg_loss(seg.reshape(-1, K), past_label.reshape(-1), pad_idx)
    res["seg_correct"], res["seg_total"] = counts(seg.reshape(-1, K), past_label.reshape(-1), pad_idx)
    total = total + l_seg
    act = out["action"].reshape(-1, K)
    tgt = trans_future_target.reshape(-1)
    ref = last_non_padding_labels(past_label, pad_idx)
    l_act = action_loss(act, tgt, pad_idx, ref, trans_future_target[:, 0])
    res["act_correct"], res["act_total"] = counts(act, tgt, pad_idx)
    total = total + l_act
    od = normalize_duration(out["duration"], dur_mask)
    td = target_dur * dur_mask
    l_dur = torch.sum((od - td) ** 2) / (torch.sum(dur_mask) if dur_den is None else dur_den)
    total = total + l_dur
    res.update(loss_seg=l_seg, loss_action=l_act, loss_dur=l_dur, loss=total)
    return res


# ----------------------------------------------------------------------------------------
# optimiser / schedule (main_darai.py:135-138)
# ----------------------------------------------------------------------------------------
def adamw_step(param, grad, m, v, step, lr, wd=5e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor semantics, in place.  step is 1-based."""
    param.mul_(1.0 - lr * wd)
    m.mul_(b1).add_(grad, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-(lr / bc1))


def warmup_cosine_lr(epoch, base_lr, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=0.0):
    """Closed form of pl_bolts 0.3.4 LinearWarmupCosineAnnealingLR (not in the container; restated from
    its published formula -- PARITY UNPINNED).  lr used DURING epoch `epoch` (0-based)."""
    if warmup_epochs > 0 and epoch < warmup_epochs:
        if warmup_epochs == 1:
            return base_lr
        return warmup_start_lr + epoch * (base_lr - warmup_start_lr) / (warmup_epochs - 1)
    if epoch == warmup_epochs:
        return base_lr
    t = epoch - warmup_epochs
    T = max_epochs - warmup_epochs
    return eta_min + 0.5 * (base_lr - eta_min) * (1.0 + math.cos(math.pi * t / T))


# ----------------------------------------------------------------------------------------
# effective rank (build-side definition; no reference code -- SURVEY F1, Appendix A.11)
# ----------------------------------------------------------------------------------------
def effective_rank(x):
    """exp(entropy(sigma / sum sigma)) (Roy & Vetterli) of a 2-D matrix, via LAPACK svdvals."""
    s = torch.linalg.svdvals(x.double())
    psum = s.sum()
    pr = s / psum
    pr = pr[pr > 0]
    return float(torch.exp(-(pr * pr.log()).sum()))


def effective_rank_torch(x):
    """Differentiable version (for checking the custom backward)."""
    s = torch.linalg.svdvals(x)
    pr = s / s.sum()
    ent = -(torch.where(pr > 0, pr * pr.clamp_min(1e-300).log(), torch.zeros_like(pr))).sum()
    return torch.exp(ent)


# ----------------------------------------------------------------------------------------
# one full training step on CPU (the timed cpu_baseline and the step-level parity checker)
# ----------------------------------------------------------------------------------------
LIVE_PREFIXES = ("input_embed.", "depth_projection.", "depth_layernorm.", "pos_embedding", "query_embed.",
                 "fuser.blocks.", "fuser.norm.", "transformer.decoder.", "fc_seg.", "fc.", "fc_len.")


def is_live(name, bn=False):
    """Parameters that receive a gradient in the reference step; everything else has
    grad=None and is skipped by AdamW.  bn: the BN-blend variant also trains fuser.alpha and the two BatchNorms."""
    if bn and name.startswith(("fuser.alpha", "fuser.bn_rgb.", "fuser.bn_depth.")):
        return True
    return name.startswith(LIVE_PREFIXES)


class CpuTrainer:
    """fwd + 3 losses + autograd bwd + AdamW over a parameter dict (reference semantics)."""

    def __init__(self, params, pad_idx, n_head=8, n_layers=1, lr=1e-3, wd=5e-3, bn_state=None, bn_training=True,
                 unsup_depth=False, n_query=8):
        """bn_state: BatchNorm buffers -> the BN-blend variant (its alpha / bn_* parameters must be in params).
        unsup_depth: the depth-as-query model (forward_unsup_depth)."""
        self.bn_state = None if bn_state is None else {k: v.clone() for k, v in bn_state.items()}
        self.bn_training = bn_training
        self.unsup_depth, self.n_query = unsup_depth, n_query
        live = (lambda k: is_live(k, bn=True)) if bn_state is not None else is_live
        if unsup_depth:
            live = lambda k: k.startswith(UNSUP_LIVE_PREFIXES)       # noqa: E731
        self.p = {k: v.clone().requires_grad_(live(k)) for k, v in params.items()}
        self.pad_idx, self.n_head, self.n_layers = pad_idx, n_head, n_layers
        self.lr, self.wd = lr, wd
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items() if v.requires_grad}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items() if v.requires_grad}
        self.t = 0

    def step(self, batch, apply=True):
        feats, depth, lab, dur, tgt = batch
        for q in self.p.values():
            q.grad = None
        if self.unsup_depth:
            out, aux = forward_unsup_depth(self.p, (feats, lab), depth, "train", self.pad_idx, self.n_head, self.n_layers,
                                           self.n_query)
        else:
            out, aux = forward(self.p, (feats, lab), depth, "train", self.pad_idx, self.n_head, self.n_layers,
                               bn_state=self.bn_state, bn_training=self.bn_training)
        res = losses(out, lab, dur, tgt, self.pad_idx)
        res["loss"].backward()
        if apply:
            self.t += 1
            with torch.no_grad():
                for k, q in self.p.items():
                    if q.grad is not None:
                        adamw_step(q, q.grad, self.m[k], self.v[k], self.t, self.lr, self.wd)
        return res, out, aux
