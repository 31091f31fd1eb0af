"""Drop-in for the reference's ``model/futr_safuser_batchnormalization.py`` -- the variant its committed entry scripts
select (main_darai.py:29, main_utkinects.py:21) and the one with the README's "selectively blends" semantics: BatchNorm
per modality, selection score = |BN gamma|, k = int(0.1 C), learnable blend alpha, no x_res (:17-107).  The FUTR around
the fuser is the token-fusion one (:110-270 differ only in the depth resolution, 160*120 by default here as at :154).

Same class names, constructor, forward signature and state_dict keys (including the BatchNorm running statistics);
parameters are holders only, the arithmetic runs in libr3d_hip.so (r3d_amd/csrc/bnfuse.hip + the shared kernels)."""
import torch
from torch import nn

from . import futr_safuser_tokenfusion as _base


class CMFuser(_base.CMFuser):
    def __init__(self, dim, depth=1, num_heads=4, mlp_ratio=4.0, qkv_bias=False):
        super().__init__(dim, depth, num_heads, mlp_ratio, qkv_bias)
        self.alpha = nn.Parameter(torch.rand(1, 1, dim))              # :32
        self.bn_rgb = nn.BatchNorm1d(dim, affine=True)                 # :36-37 (holders: weight, bias, running statistics)
        self.bn_depth = nn.BatchNorm1d(dim, affine=True)


class FUTR(_base.FUTR):
    _fuser_cls = CMFuser

    def __init__(self, n_class, hidden_dim, src_pad_idx, device, args, n_query=8, n_head=8, num_encoder_layers=6,
                 num_decoder_layers=6, query_num=49, depth_pixels=160 * 120):
        super().__init__(n_class, hidden_dim, src_pad_idx, device, args, n_query, n_head, num_encoder_layers,
                         num_decoder_layers, query_num, depth_pixels)
