"""ctypes binding of libr3d_hip.so (include/r3d_hip.h).  There is NO fallback: if the library is missing or stale
the import of the compute path fails loudly -- the product never runs on a CPU/PyTorch substitute."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libr3d_hip.so")
ABI_VERSION = 2

c_float_p = C.c_void_p      # raw device addresses travel as integers (tensor.data_ptr())


class GemmDesc(C.Structure):
    """struct r3d_gemm_desc (include/r3d_hip.h)."""
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("layout", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("a_add", C.c_void_p), ("a_add_mod", C.c_int32), ("a_add_ld", C.c_int32), ("a_row_xor", C.c_int32),
        ("b_add", C.c_void_p), ("b_add_mod", C.c_int32), ("b_add_ld", C.c_int32),
        ("bias", C.c_void_p),
        ("pre_out", C.c_void_p), ("ldpre", C.c_int32),
        ("act", C.c_int32),
        ("drop_mask", C.c_void_p), ("lddrop", C.c_int32), ("drop_scale", C.c_float),
        ("aux", C.c_void_p), ("ldaux", C.c_int32), ("mul", C.c_int32),
        ("res1", C.c_void_p), ("ldr1", C.c_int32),
        ("res2", C.c_void_p), ("ldr2", C.c_int32),
        ("alpha", C.c_float), ("accumulate", C.c_int32),
        ("bias_grad", C.c_void_p),
        ("c_row_xor", C.c_int32),
        ("splitk", C.c_int32), ("k_per_split", C.c_int32), ("partial", C.c_void_p),
        ("tile", C.c_int32),
        ("vec", C.c_int32),
        ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_lr", C.c_void_p), ("adam_step", C.c_void_p),
        ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float), ("adam_wd", C.c_float),
        ("adam_gscale", C.c_float),
        ("prec", C.c_int32),
    ]


class LnFwdJob(C.Structure):
    """struct r3d_ln_fwd_job"""
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int32), ("nsplit", C.c_int32), ("bias", C.c_void_p), ("pre_out", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("y", C.c_void_p), ("ldy", C.c_int32), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("pair_out", C.c_void_p), ("rows", C.c_int32), ("H", C.c_int32), ("relu", C.c_int32)]


class LnBwdJob(C.Structure):
    """struct r3d_ln_bwd_job"""
    _fields_ = [("dy", C.c_void_p), ("lddy", C.c_int32), ("pair_in", C.c_int32), ("dy2", C.c_void_p), ("lddy2", C.c_int32),
                ("x", C.c_void_p), ("ldx", C.c_int32), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("relu", C.c_int32), ("add1", C.c_void_p), ("ldadd1", C.c_int32), ("add2", C.c_void_p),
                ("ldadd2", C.c_int32), ("dx", C.c_void_p), ("lddx", C.c_int32), ("dx2", C.c_void_p), ("lddx2", C.c_int32),
                ("drop_mask", C.c_void_p), ("lddrop", C.c_int32), ("drop_scale", C.c_float), ("dgamma", C.c_void_p),
                ("dbeta", C.c_void_p), ("ws", C.c_void_p), ("rows", C.c_int32), ("H", C.c_int32),
                ("rows_per_block", C.c_int32), ("nblocks", C.c_int32)]


class GemmLnJob(C.Structure):
    """struct r3d_gemm_ln_job"""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int32), ("W", C.c_void_p), ("ldw", C.c_int32), ("bias", C.c_void_p),
                ("drop_mask", C.c_void_p), ("lddrop", C.c_int32), ("drop_scale", C.c_float),
                ("res1", C.c_void_p), ("ldr1", C.c_int32), ("res2", C.c_void_p), ("ldr2", C.c_int32),
                ("pre_out", C.c_void_p), ("ldpre", C.c_int32), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("y", C.c_void_p), ("ldy", C.c_int32), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("pair_out", C.c_void_p), ("M", C.c_int32), ("K", C.c_int32)]


class FuserChainFwdArgs(C.Structure):
    """struct r3d_fuser_chain_fwd_args"""
    _PTRS = ("x0 h1 wv wproj bproj g2 be2 w1 b1 w2 b2 gf bef pos wkv bkv wseg bseg vsw x1 h2 m2 r2 u f1 x3 y mf rf fused seg "
             "cakv qpos w_in b_in w_out b_out g1 be1 wq bq drop_sa drop_d1").split()
    _PTRS2 = "sa_qkv p_sa sa_o t1_pre t1 m1 r1 caq".split()
    _PLANES = "pl_wv pl_wproj pl_w1 pl_w2 pl_wkv pl_wseg".split()
    _fields_ = ([(n, C.c_void_p) for n in _PTRS] + [("drop_scale", C.c_float)] + [(n, C.c_void_p) for n in _PTRS2] +
                [(n, C.c_int32) for n in "N S K H add_xres B Q heads".split()] + [("timeline", C.c_void_p)] +
                [(n, C.c_void_p) for n in _PLANES])


class PlaneJob(C.Structure):
    """struct r3d_plane_job"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("ld", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("transposed", C.c_int32), ("first_block", C.c_int32), ("pad_", C.c_int32)]


class FuserChainBwdArgs(C.Structure):
    """struct r3d_fuser_chain_bwd_args"""
    _PTRS = ("d_cakv d_seg d_extra wkv wseg x3 mf rf gf w2 u w1 x1 m2 r2 g2 wproj wv x0 m1 r1 g1n drop_x0 m_rgb m_dep rgb "
             "dep_pre mean_d rstd_d lnd_g lnd_b d_fused d_x3 d_u d_h2 d_x1 d_v d_h1 d_rgb_pre d_dep_pre part_nf part_n2 "
             "part_n1 part_dep d_caq d_t1_res wq t1_pre m1d r1d g1d drop_d1 w_out sa_qkv p_sa drop_sa w_in caqin t1pre_out "
             "sap sao saqkv sain part_d1").split()
    _PLANES = "pl_wkv_t pl_wseg_t pl_w2_t pl_w1_t pl_wproj_t pl_wv_t".split()
    _fields_ = ([(n, C.c_void_p) for n in _PTRS] + [("drop_scale", C.c_float)] +
                [(n, C.c_int32) for n in "N S K H add_xres B Q heads".split()] + [("timeline", C.c_void_p)] +
                [(n, C.c_void_p) for n in _PLANES])


class DecoderChainArgs(C.Structure):
    """struct r3d_decoder_chain_args"""
    _PTRS = ("caq cakv key_label p_ca drop_ca ca_o wo bo drop_d2 t1 t2_pre g2 be2 t2 m2 r2 w1 b1 drop_ff ff1 w2 b2 drop_d3 "
             "t3_pre d_t3pre d_ff2 d_ff1 d_t2pre d_cap d_cao d_caq d_cakv part_d2").split()
    _fields_ = ([(n, C.c_void_p) for n in _PTRS] + [("drop_scale", C.c_float)] +
                [(n, C.c_int32) for n in "pad_idx B S H Q heads phases".split()] + [("timeline", C.c_void_p)] +
                [(n, C.c_void_p) for n in "pl_wo pl_w1 pl_w2 pl_w2_t pl_w1_t pl_wo_t".split()])


class MhaJob(C.Structure):
    """struct r3d_mha_job"""
    _fields_ = [("q", C.c_void_p), ("ldq", C.c_int32), ("k", C.c_void_p), ("ldk", C.c_int32), ("v", C.c_void_p),
                ("ldv", C.c_int32), ("key_padding_mask", C.c_void_p), ("key_label", C.c_void_p), ("pad_idx", C.c_int32),
                ("probs", C.c_void_p), ("drop_mask", C.c_void_p), ("drop_scale", C.c_float), ("o", C.c_void_p),
                ("ldo", C.c_int32), ("B", C.c_int32), ("heads", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32),
                ("dh", C.c_int32)]


class MhaBwdJob(C.Structure):
    """struct r3d_mha_bwd_job"""
    _fields_ = [("q", C.c_void_p), ("ldq", C.c_int32), ("k", C.c_void_p), ("ldk", C.c_int32), ("v", C.c_void_p),
                ("ldv", C.c_int32), ("probs", C.c_void_p), ("drop_mask", C.c_void_p), ("drop_scale", C.c_float),
                ("d_o", C.c_void_p), ("lddo", C.c_int32), ("dq", C.c_void_p), ("lddq", C.c_int32), ("dk", C.c_void_p),
                ("lddk", C.c_int32), ("dv", C.c_void_p), ("lddv", C.c_int32), ("B", C.c_int32), ("heads", C.c_int32),
                ("Lq", C.c_int32), ("Lk", C.c_int32), ("dh", C.c_int32)]


class RowsumJob(C.Structure):
    """struct r3d_rowsum_job"""
    _fields_ = [("src1", C.c_void_p), ("src2", C.c_void_p), ("dst", C.c_void_p), ("ld1", C.c_int32), ("ld2", C.c_int32),
                ("ldd", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("mod", C.c_int32)]


class TailLossesArgs(C.Structure):
    """r3d_tail_losses_args (include/r3d_hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("x", "g3", "b3", "gF", "bF", "w_head", "b_head")] + [("n_head", C.c_int32)] +
                [(n, C.c_void_p) for n in ("t3", "m3", "r3", "tgtF", "mF", "rF", "out")] +
                [("ld_out", C.c_int32), ("H", C.c_int32), ("seg", C.c_void_p), ("ld_seg", C.c_int32),
                 ("past_label", C.c_void_p), ("target", C.c_void_p), ("target_dur", C.c_void_p)] +
                [(n, C.c_int32) for n in ("B", "S", "Q", "K", "pad_idx", "exclude_idx")] +
                [("dur_den", C.c_void_p), ("grad_scale", C.c_float), ("d_seg", C.c_void_p), ("ld_dseg", C.c_int32),
                 ("d_out", C.c_void_p), ("ld_dout", C.c_int32), ("loss_out", C.c_void_p), ("counts", C.c_void_p),
                 ("tick_a", C.c_void_p), ("tick_b", C.c_void_p), ("drop", C.c_void_p), ("drop_scale", C.c_float)] +
                [(n, C.c_void_p) for n in ("dx", "dx2", "wsF", "ws3")] + [("defer_finalize", C.c_int32)])


class LossFinalizeJob(C.Structure):
    """struct r3d_loss_finalize_job"""
    _fields_ = [("part", C.c_void_p), ("B", C.c_int32), ("S", C.c_int32), ("Q", C.c_int32), ("has_seg", C.c_int32),
                ("dur_den", C.c_void_p), ("loss_out", C.c_void_p), ("counts", C.c_void_p), ("acc_loss", C.c_void_p),
                ("acc_counts", C.c_void_p)]


class LnFinalizeJob(C.Structure):
    """struct r3d_ln_finalize_job"""
    _fields_ = [("ws", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("rows", C.c_int32), ("H", C.c_int32)]


GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2

_I, _L, _F, _P, _D = C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_double

_SIGNATURES = {
    "r3d_abi_version": ([], C.c_int),
    "r3d_build_info": ([C.c_char_p, _I], C.c_int),
    "r3d_allreduce_flat": ([_P, _L, _P, _P], C.c_int),
    "r3d_gemm_f32": ([C.POINTER(GemmDesc), _P], C.c_int),
    "r3d_splitk_reduce": ([C.POINTER(GemmDesc), _P], C.c_int),
    "r3d_gemm_partial_floats": ([C.c_int32, C.c_int32, C.c_int32], C.c_int64),
    "r3d_gemm_plan": ([C.POINTER(GemmDesc)], C.c_int),
    "r3d_gemm_grouped_prepare": ([C.POINTER(GemmDesc), _I, _I, C.POINTER(C.c_int32)], C.c_int),
    "r3d_gemm_grouped_launch": ([_P, _P, C.POINTER(C.c_int32), _I, _I, _I, _I, _P], C.c_int),
    "r3d_gemm_ln_supported": ([_I, _I, _I], C.c_int),
    "r3d_gemm_ln_fwd": ([_P, _I, _I, _P], C.c_int),
    "r3d_fuser_chain_supported": ([_I, _I, _I, _I, _I, _I], C.c_int),
    "r3d_fuser_chain_fwd": ([_P, _P], C.c_int),
    "r3d_fuser_chain_bwd": ([_P, _P], C.c_int),
    "r3d_weight_plane_elems": ([_I, _I], C.c_int64),
    "r3d_weight_planes": ([_P, _I, _I, _P], C.c_int),
    "r3d_token_exchange3_fwd": ([_P, _P, _P, _P, _P, _F, _P, _I, _I, _P], C.c_int),
    "r3d_token_exchange3_bwd": ([_P, _P, _P, _F, _P, _P, _P, _I, _I, _P], C.c_int),
    "r3d_attn3_fwd": ([_P, _P, _P, _I, _I, _I, _P], C.c_int),
    "r3d_attn3_bwd": ([_P, _P, _P, _I, _I, _I, _P], C.c_int),
    "r3d_triple_mean_fwd": ([_P, _P, _I, _I, _P], C.c_int),
    "r3d_triple_mean_bwd": ([_P, _P, _I, _I, _P], C.c_int),
    "r3d_decoder_chain_supported": ([_I, _I, _I, _I], C.c_int),
    "r3d_decoder_chain": ([_P, _P, _P, _P], C.c_int),
    "r3d_gemm_ln_mha_supported": ([_I, _I, _I, _I], C.c_int),
    "r3d_gemm_ln_mha_fwd": ([_P, _I, _I, _P, _P], C.c_int),
    "r3d_layernorm_fwd_multi": ([_P, _I, _P], C.c_int),
    "r3d_layernorm_bwd_multi": ([_P, _I, _P], C.c_int),
    "r3d_layernorm_bwd_multi_mha": ([_P, _I, _P, _P], C.c_int),
    "r3d_layernorm_bwd_finalize_batched": ([_P, _I, _I, _P], C.c_int),
    "r3d_layernorm_fwd": ([_P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P], C.c_int),
    "r3d_layernorm_bwd_ws_floats": ([_I, _I], C.c_int64),
    "r3d_add_rowbcast": ([_P, _I, _P, _I, _I, _P, _I, _I, _I, _P], C.c_int),
    "r3d_layernorm_bwd": ([_P, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _F,
                           _P, _P, _P, _I, _I, _I, _P], C.c_int),
    "r3d_layernorm_bwd_finalize": ([_P, _I, _I, _P, _P, _P], C.c_int),
    "r3d_colsum_ws_floats": ([_I, _I], C.c_int64),
    "r3d_colsum": ([_P, _I, _I, _I, _P, _P, _I, _P], C.c_int),
    "r3d_rowmod_sum": ([_P, _I, _I, _I, _I, _P, _I, _I, _P], C.c_int),
    "r3d_rowmod_sum_batched": ([_P, _I, _I, _I, _P], C.c_int),
    "r3d_tick": ([_P, _P, _P], C.c_int),
    "r3d_colabssum": ([_P, _I, _I, _I, _P, _P], C.c_int),
    "r3d_token_select": ([_P, _P, _D, _I, _I, _I, _P, _P, _P, _P], C.c_int),
    "r3d_token_exchange_fwd": ([_P, _P, _P, _P, _P, _P, _F, _I, _I, _P], C.c_int),
    "r3d_token_exchange_bwd": ([_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P], C.c_int),
    "r3d_mha_core_fwd": ([_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "r3d_mha_core_bwd": ([_P, _I, _P, _I, _P, _I, _P, _P, _F, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
                         C.c_int),
    "r3d_decoder_fused_supported": ([_I, _I, _I, _I], C.c_int),
    "r3d_decoder_layer_fwd": ([_P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P], C.c_int),
    "r3d_embed_fuse_fwd": ([_P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I,
                            _P], C.c_int),
    "r3d_gemm_bf3_nt_pair": ([_P, _P, _P], C.c_int),
    "r3d_embed_fuse_fwd_planes": ([_P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I,
                                   _I, _P, _I, _I, _P], C.c_int),
    "r3d_embed_fuse_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
                           C.c_int),
    "r3d_bn_stats": ([_P] * 13 + [_I, _I, _I, _F, _P], C.c_int),
    "r3d_bn_blend_fwd": ([_P] * 12 + [_F] + [_P] * 6 + [_I, _I, _P], C.c_int),
    "r3d_bn_blend_bwd": ([_P] * 7 + [_F] + [_P] * 17 + [_I, _I, _P], C.c_int),
    "r3d_bn_bwd_apply": ([_P] * 14 + [_I, _I, _I, _P], C.c_int),
    "r3d_bn_sync_pack": ([_P, _P, _I, _I, _P, _P], C.c_int),
    "r3d_bn_sync_finalize": ([_P, _I, _I, _I] + [_P] * 9 + [_F, _P], C.c_int),
    "r3d_decoder_tail_fwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P], C.c_int),
    "r3d_decoder_tail_bwd": ([_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I,
                              _P], C.c_int),
    "r3d_decoder_tail_losses_supported": ([_I, _I, _I, _I], C.c_int),
    "r3d_decoder_tail_losses": ([_P, _P, _P], C.c_int),
    "r3d_losses_fwd_bwd": ([_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _F, _P, _I, _P, _I,
                            _P, _I, _P, _P, _P, _P, _P, _P], C.c_int),
    "r3d_losses_ws_floats": ([_I, _I, _I], C.c_int64),
    "r3d_adamw_flat": ([_P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _F, _P], C.c_int),
    "r3d_adamw_flat_dropout": ([_P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _F, _P, _L, _F, C.c_uint64, _P, _P], C.c_int),
    "r3d_losses_finalize": ([_P, _P], C.c_int),
    "r3d_adamw_flat_fin": ([_P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _F, _P, _P], C.c_int),
    "r3d_adamw_flat_dropout_fin": ([_P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _F, _P, _L, _F, C.c_uint64, _P, _P, _P], C.c_int),
    "r3d_adamw_2d": ([_P, _P, _P, _P, _I, _I, _I, _P, _P, _F, _F, _F, _F, _F, _P], C.c_int),
    "r3d_dropout_mask": ([_P, _L, _F, C.c_uint64, _P, _P], C.c_int),
    "r3d_erank_lds_bytes": ([_I, _I], C.c_int64),
    "r3d_erank_jacobi": ([_P, _I, _L, _I, _I, _I, _I, _P, _P, _P, _I, _P], C.c_int),
    "r3d_erank_blocked_sizes": ([_I, _I, _I, _P], C.c_int),
    "r3d_erank_blocked": ([_P, _I, _I, _I, _P, _P, _P, _P, _I, _P], C.c_int),
    "r3d_erank_blocked_t": ([_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P], C.c_int),
    "r3d_erank_bwd_coef": ([_P, _P, _P, _P, _I, _I, _P], C.c_int),
    "r3d_erank_bwd_coef2": ([_P, _P, _P, _P, _P, _I, _I, _P], C.c_int),
    "r3d_erank_bwd_fix": ([_P, _P, _P, _I, _I, _P], C.c_int),
    "r3d_scale_rows": ([_P, _I, _I, _I, _P, _P], C.c_int),
    "r3d_posenc_fwd": ([_P, _I, _P, _I, _I, _P, _F, _P, _I, _I, _I, _P], C.c_int),
    "r3d_posenc_bwd": ([_P, _I, _P, _F, _P, _I, _P, _I, _I, _I, _P], C.c_int),
    "r3d_avgpool_rows_fwd": ([_P, _I, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "r3d_avgpool_rows_bwd": ([_P, _I, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "r3d_embed_gather_fwd": ([_P, _I, _P, _P, _I, _I, _P, _I, _I, _I, _P], C.c_int),
    "r3d_embed_gather_bwd": ([_P, _I, _P, _P, _I, _I, _I, _P], C.c_int),
    "r3d_erank_lds_bytes_v": ([_I, _I], C.c_int64),
    "r3d_erank_jacobi_warm": ([_P, _I, _L, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P], C.c_int),
    "r3d_erank_vt_polish": ([_P, _P, _P, _L, _P], C.c_int),
}

EXPORTS = tuple(_SIGNATURES)

_lib = None


class R3DHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library.  Raises R3DHipError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise R3DHipError(
            f"{LIB_PATH} is missing: the HIP kernels are the only compute path of r3d_amd (no CPU/PyTorch fallback). "
            f"Build it with `python -m r3d_amd.build` (needs hipcc; cross-compiles for gfx950 without a GPU).")
    lib = C.CDLL(LIB_PATH)
    for name, (argt, rest) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise R3DHipError(f"{LIB_PATH} does not export {name}: stale build, run `python -m r3d_amd.build --force`") from e
        fn.argtypes = argt
        fn.restype = rest
    v = lib.r3d_abi_version()
    if v != ABI_VERSION:
        raise R3DHipError(f"libr3d_hip.so ABI {v} != expected {ABI_VERSION}: rebuild with `python -m r3d_amd.build --force`")
    _lib = lib
    return lib


_ERR = {-1: "R3D_EINVAL (rejected argument)", -2: "R3D_EALIGN (misaligned pointer / leading dimension)",
        -3: "R3D_ENORCCL (no RCCL mapped into the process)"}


def check(rc, what):
    if rc <= -100:
        raise R3DHipError(f"{what} failed: ncclResult_t {-100 - rc} from RCCL")
    if rc != 0:
        raise R3DHipError(f"{what} failed: {_ERR.get(rc, f'hipError_t {rc}' if rc > 0 else rc)}")


def build_info():
    buf = C.create_string_buffer(128)
    check(load().r3d_build_info(buf, 128), "r3d_build_info")
    return buf.value.decode()
