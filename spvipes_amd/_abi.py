"""ctypes binding of libspvipes_hip.so (C ABI: include/spvipes_hip.h).

There is NO fallback: if the library is missing or a call fails, this raises.  Tensors are passed
as raw device pointers; every call is enqueued on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libspvipes_hip.so")

SPV_COUNT_F32, SPV_COUNT_U16 = 0, 1
NB_CMAX, DEC_KP, DEC_KS, DEC_CELLS_PER_WG = 64, 16, 32, 128
DEC_KPS = DEC_KP + DEC_KS


class SpvCounts(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ld", C.c_int64), ("rows", C.c_void_p), ("col_off", C.c_int32), ("dtype", C.c_int32)]


class SpvDecParams(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("ldx", C.c_int64), ("rows", C.c_void_p), ("col_off", C.c_int32), ("count_is_u16", C.c_int32),
        ("B", C.c_int32), ("G", C.c_int32), ("Bp", C.c_int32), ("Gp", C.c_int32),
        ("logits", C.c_void_p), ("n_gene_tiles", C.c_int32), ("logits_f32", C.c_int32),
        ("Wps_hi", C.c_void_p), ("Wps_lo", C.c_void_p), ("Aps_hi", C.c_void_p), ("Aps_lo", C.c_void_p),
        ("gene_tab", C.c_void_p), ("cnt_tab", C.c_void_p),
        ("a_p", C.c_void_p), ("a_s", C.c_void_p), ("lse_p", C.c_void_p), ("lse_s", C.c_void_p), ("w_row", C.c_void_p),
        ("gene_splits", C.c_int32), ("genes_per_split", C.c_int32),
        ("part_max_p", C.c_void_p), ("part_sum_p", C.c_void_p), ("part_max_s", C.c_void_p), ("part_sum_s", C.c_void_p),
        ("rec_part", C.c_void_p), ("tp_part", C.c_void_p), ("ts_part", C.c_void_p), ("dtheta_part", C.c_void_p),
        ("dL", C.c_void_p), ("tP", C.c_void_p), ("tS", C.c_void_p), ("grads_f32", C.c_int32),
        ("nb_splits", C.c_int32), ("nb_genes_per_split", C.c_int32), ("nb_cell_tiles", C.c_int32),
    ]


class SpvReduceProb(C.Structure):
    _fields_ = [("src", C.c_void_p), ("slab_stride", C.c_int64), ("ld_src", C.c_int64), ("nslabs", C.c_int32), ("col_off", C.c_int32),
                ("rows", C.c_int32), ("cols", C.c_int32), ("dst", C.c_void_p), ("ld_dst", C.c_int64), ("accumulate", C.c_int32),
                ("pad_", C.c_int32), ("alpha", C.c_void_p), ("exp_scale", C.c_void_p)]


SPV_MAXR = 16


class SpvReduceBatch(C.Structure):
    _fields_ = [("p", SpvReduceProb * SPV_MAXR), ("nprob", C.c_int32)]


SPV_MAXP = 8
POE_LMAX = 1024  # label codes the pairing kernel accepts: integers in [0, POE_LMAX)
BN_ROWS = 32  # rows per workgroup of the BatchNorm kernels (sizes their partial-sum workspace)


class SpvLinearProb(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("W", C.c_void_p), ("bias", C.c_void_p), ("Y", C.c_void_p), ("ldy", C.c_int64),
                ("dY", C.c_void_p), ("lddy", C.c_int64), ("dX", C.c_void_p), ("lddx", C.c_int64), ("dW", C.c_void_p), ("db", C.c_void_p),
                ("N", C.c_int32), ("K", C.c_int32), ("keep", C.c_void_p), ("W2", C.c_void_p), ("n_w2", C.c_int32),
                ("img_hi", C.c_void_p), ("img_lo", C.c_void_p), ("ld_img", C.c_int64), ("img_rows", C.c_int32)]


class SpvLinearBatch(C.Structure):
    _fields_ = [("p", SpvLinearProb * SPV_MAXP), ("nprob", C.c_int32), ("B", C.c_int32), ("relu", C.c_int32), ("drop_p", C.c_float),
                ("seed", C.c_uint64), ("seed_ptr", C.c_void_p), ("accumulate", C.c_int32)]


class SpvBnProb(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("Y", C.c_void_p), ("ldy", C.c_int64), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("stats", C.c_void_p), ("part", C.c_void_p),
                ("dY", C.c_void_p), ("lddy", C.c_int64), ("dX", C.c_void_p), ("lddx", C.c_int64), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("N", C.c_int32), ("img_hi", C.c_void_p), ("img_lo", C.c_void_p), ("ld_img", C.c_int64), ("img_rows", C.c_int32)]


class SpvBnBatch(C.Structure):
    _fields_ = [("p", SpvBnProb * SPV_MAXP), ("nprob", C.c_int32), ("B", C.c_int32), ("training", C.c_int32), ("relu", C.c_int32),
                ("eps", C.c_float), ("momentum", C.c_float)]


class SpvSampleProb(C.Structure):
    _fields_ = [("post", C.c_void_p), ("n", C.c_int32), ("eps", C.c_void_p), ("scale", C.c_void_p), ("logz", C.c_void_p), ("theta", C.c_void_p),
                ("kl", C.c_void_p), ("g_loc", C.c_void_p), ("g_logvar", C.c_void_p), ("g_scale", C.c_void_p), ("g_logz", C.c_void_p),
                ("g_kl", C.c_void_p), ("d_post", C.c_void_p), ("g_ld", C.c_int64)]


class SpvSampleBatch(C.Structure):
    _fields_ = [("p", SpvSampleProb * SPV_MAXP), ("nprob", C.c_int32), ("B", C.c_int32)]


class SpvPoeArgs(C.Structure):
    _fields_ = [("stats", C.c_void_p * 2), ("ld", C.c_int64 * 2), ("partner", C.c_void_p * 2), ("mode", C.c_void_p * 2), ("eps", C.c_void_p * 2),
                ("loc", C.c_void_p * 2), ("logvar", C.c_void_p * 2), ("scale", C.c_void_p * 2), ("logz", C.c_void_p * 2), ("theta", C.c_void_p * 2),
                ("kl", C.c_void_p * 2), ("g_loc", C.c_void_p * 2), ("g_logvar", C.c_void_p * 2), ("g_scale", C.c_void_p * 2),
                ("g_logz", C.c_void_p * 2), ("g_kl", C.c_void_p * 2), ("d_stats", C.c_void_p * 2), ("B", C.c_int32 * 2), ("n", C.c_int32),
                ("clamp_scale", C.c_int32), ("lone_passthrough", C.c_int32), ("expert", C.c_void_p * 2), ("ld_expert", C.c_int64 * 2),
                ("d_expert", C.c_void_p * 2), ("lab", C.c_void_p * 2), ("order", C.c_void_p * 2), ("rank", C.c_void_p * 2), ("tables", C.c_void_p)]


SPV_POE_MAXG, POE_COMP_SEG, POE_COMP_CMAX = 4, 16, 64


class SpvFc1FwdArgs(C.Structure):
    _fields_ = [("x", C.POINTER(SpvCounts)), ("B", C.c_int32), ("G", C.c_int32), ("W1_hi", C.c_void_p), ("W1_lo", C.c_void_p), ("ldw", C.c_int64),
                ("N1", C.c_int32), ("bias", C.c_void_p), ("bias2", C.c_void_p), ("n_first", C.c_int32), ("nsplit", C.c_int32), ("splits", C.c_int32),
                ("slabs", C.c_void_p), ("rowsum_ws", C.c_void_p), ("h1", C.c_void_p), ("library", C.c_void_p), ("xb_all", C.c_void_p),
                ("ld_xb", C.c_int64), ("library_all", C.c_void_p), ("cov", C.c_void_p), ("cov_idx", C.c_void_p)]


class SpvFc1BwdArgs(C.Structure):
    _fields_ = [("dh1", C.c_void_p), ("h1", C.c_void_p), ("x", C.POINTER(SpvCounts)), ("B", C.c_int32), ("G", C.c_int32), ("N1", C.c_int32),
                ("n_first", C.c_int32), ("nsplit", C.c_int32), ("Bp", C.c_int32), ("dh_hi", C.c_void_p), ("dh_lo", C.c_void_p), ("ld_dh", C.c_int64),
                ("part", C.c_void_p), ("db", C.c_void_p), ("db2", C.c_void_p), ("dW", C.c_void_p), ("dW2", C.c_void_p), ("ldc", C.c_int64),
                ("xb", C.c_void_p), ("ld_xb", C.c_int64), ("scale_ws", C.c_void_p)]


class SpvGatherProb(C.Structure):
    _fields_ = [("src", C.c_void_p), ("idx", C.c_void_p), ("dst", C.c_void_p), ("count", C.c_int64)]


def gather_u32(probs) -> None:
    """[(src, idx or None, dst)] -> dst[i] = src[idx[i]] (or src[i]) on 4-byte words, one launch for all problems (spv_gather_u32)"""
    arr = (SpvGatherProb * len(probs))()
    for a, (src, idx, dst) in zip(arr, probs):
        a.src, a.idx, a.dst, a.count = ptr(src), ptr(idx), ptr(dst), dst.numel()
    call("spv_gather_u32", arr, len(probs), stream_ptr())


ADAM_MAX_IMAGES = 16
IMAGE_BF16, IMAGE_F16 = 0, 1   # spv_adam_image.fmt


class SpvAdamImage(C.Structure):
    _fields_ = [("begin", C.c_int64), ("count", C.c_int64), ("cols", C.c_int32), ("row_off", C.c_int32), ("col_off", C.c_int32), ("fmt", C.c_int32),
                ("ld", C.c_int64), ("dst", C.c_void_p), ("scale", C.c_float), ("_pad", C.c_int32)]


class SpvPoeCompArgs(C.Structure):
    _fields_ = ([("ngroups", C.c_int32), ("n", C.c_int32), ("ncomp", C.c_int32), ("pad_", C.c_int32), ("B", C.c_int32 * SPV_POE_MAXG),
                 ("stats", C.c_void_p * SPV_POE_MAXG), ("ld", C.c_int64 * SPV_POE_MAXG), ("comp", C.c_void_p * SPV_POE_MAXG),
                 ("eps", C.c_void_p * SPV_POE_MAXG), ("part", C.c_void_p), ("mean", C.c_void_p)]
                + [(k, C.c_void_p * SPV_POE_MAXG) for k in ("loc", "logvar", "scale", "logz", "theta", "kl", "g_loc", "g_logvar", "g_scale", "g_logz", "g_kl",
                                                            "dpn", "d_stats")])


class SpvPlan(C.Structure):
    _fields_ = [("ptr0", C.c_void_p), ("ind0", C.c_void_p), ("val0", C.c_void_p), ("n0", C.c_int32),
                ("ptr1", C.c_void_p), ("ind1", C.c_void_p), ("val1", C.c_void_p), ("n1", C.c_int32)]


class SpvPlanExpertArgs(C.Structure):
    _fields_ = [("plan", SpvPlan), ("idx", C.c_void_p * 2), ("inv", C.c_void_p * 2), ("comp", C.c_void_p * 2), ("stats", C.c_void_p * 2),
                ("ld", C.c_int64 * 2), ("expert", C.c_void_p * 2), ("ld_expert", C.c_int64 * 2), ("rowsum", C.c_void_p * 2),
                ("d_expert", C.c_void_p * 2), ("d_stats", C.c_void_p * 2), ("B", C.c_int32), ("n", C.c_int32)]


class SpvZsplitArgs(C.Structure):
    _fields_ = [("priv", C.c_void_p * 2), ("poe", C.c_void_p * 2), ("zcat", C.c_void_p * 2), ("d_zcat", C.c_void_p * 2),
                ("d_priv", C.c_void_p * 2), ("d_poe", C.c_void_p * 2), ("B", C.c_int32), ("n_p", C.c_int32), ("n_s", C.c_int32), ("ngroups", C.c_int32),
                ("am_hi", C.c_void_p * 2), ("am_lo", C.c_void_p * 2), ("ld_am", C.c_int64), ("am_col", C.c_int32), ("am_cols", C.c_int32),
                ("aps_hi", C.c_void_p * 2), ("aps_lo", C.c_void_p * 2), ("Bp", C.c_int32)]


class SpvFoldProb(C.Structure):
    _fields_ = [("W", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("zsum", C.c_void_p), ("zz", C.c_void_p), ("z", C.c_void_p), ("ldz", C.c_int64), ("stat", C.c_void_p),
                ("img_hi", C.c_void_p), ("img_lo", C.c_void_p), ("ld_img", C.c_int64), ("col_off", C.c_int32), ("slot", C.c_int32),
                ("dWeff", C.c_void_p), ("ld_dw", C.c_int64), ("dW", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("red_part", C.c_void_p), ("dz", C.c_void_p), ("lddz", C.c_int64), ("G", C.c_int32), ("Gp", C.c_int32), ("K", C.c_int32),
                ("zcol", C.c_int32), ("out_priv", C.c_void_p), ("out_poe", C.c_void_p), ("n_p", C.c_int32), ("n_s", C.c_int32)]


class SpvFoldBatch(C.Structure):
    _fields_ = [("p", SpvFoldProb * SPV_MAXP), ("nprob", C.c_int32), ("B", C.c_int32), ("training", C.c_int32), ("eps", C.c_float),
                ("momentum", C.c_float)]


class SpvGemmFixup(C.Structure):
    _fields_ = [("counters", C.c_void_p), ("alpha", C.c_void_p), ("dst0", C.c_void_p), ("ld0", C.c_int64), ("n0", C.c_int32),
                ("dst1", C.c_void_p), ("ld1", C.c_int64), ("c1", C.c_int32), ("n1", C.c_int32)]


class SpvGemmArgs(C.Structure):
    """one group's arguments of spv_gemm_bf16, for spv_gemm_bf16_grouped"""
    _fields_ = [("a_kmajor", C.c_int32), ("pad0_", C.c_int32), ("A_hi", C.c_void_p), ("A_lo", C.c_void_p), ("lda", C.c_int64),
                ("B_hi", C.c_void_p), ("B_lo", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("nsplit", C.c_int32), ("splits", C.c_int32), ("a_tiles", C.c_int32),
                ("slab_stride", C.c_int64)]


class SpvDecGroup(C.Structure):
    """one group's decoder launch arguments, for the spv_dec_*_grouped entry points (include/spvipes_hip.h: spv_dec_group)"""
    _fields_ = [("p", SpvDecParams), ("library", C.c_void_p), ("px_r", C.c_void_p),
                ("Am_hi", C.c_void_p), ("Am_lo", C.c_void_p), ("Wm_hi", C.c_void_p), ("Wm_lo", C.c_void_p), ("K", C.c_int32), ("nsplit", C.c_int32),
                ("Tp", C.c_void_p), ("Ts", C.c_void_p), ("dz_part", C.c_void_p), ("dw_part", C.c_void_p)]


TRUNK_KMAX = 48


class SpvTrunkProb(C.Structure):
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("zsum", C.c_void_p), ("zz", C.c_void_p), ("Wf", C.c_void_p), ("cf", C.c_void_p), ("stat", C.c_void_p),
                ("dWf", C.c_void_p), ("dcf", C.c_void_p), ("dW", C.c_void_p), ("dbias", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("dred", C.c_void_p), ("z", C.c_void_p), ("ldz", C.c_int64), ("dz", C.c_void_p), ("lddz", C.c_int64), ("N", C.c_int32), ("K", C.c_int32)]


class SpvTrunkBatch(C.Structure):
    _fields_ = [("p", SpvTrunkProb * 2), ("nprob", C.c_int32), ("B", C.c_int32), ("training", C.c_int32), ("eps", C.c_float), ("momentum", C.c_float)]


_SIGNATURES = {
    "spv_version": (C.c_int, []),
    "spv_last_error": (C.c_char_p, []),
    "spv_build_id": (C.c_char_p, []),
    "spv_pack_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "spv_fc1_w_scale": (C.c_float, []),
    "spv_pack_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "spv_enc_fc1_fwd_uses_dma": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64]),
    "spv_enc_fc1_fwd": (C.c_int, [C.POINTER(SpvCounts), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spv_enc_fc1_wgrad_split_uses_dma": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64]),
    "spv_enc_fc1_fwd_grouped": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "spv_enc_fc1_bwd_grouped": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "spv_prepare_log1p": (C.c_int, [C.POINTER(SpvCounts), C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "spv_prepare_log1p_split": (C.c_int, [C.POINTER(SpvCounts), C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "spv_enc_fc1_wgrad": (C.c_int, [C.POINTER(SpvCounts), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "spv_enc_fc1_bwd_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "spv_dec_heads_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spv_gemm_bf16_uses_dma": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "spv_gemm_bf16": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "spv_gemm_bf16_fix": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(SpvGemmFixup), C.c_void_p]),
    "spv_dec_tables": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spv_dec_lse": (C.c_int, [C.POINTER(SpvDecParams), C.c_void_p, C.c_void_p]),
    "spv_dec_nb_fwd": (C.c_int, [C.POINTER(SpvDecParams), C.c_int32, C.c_void_p]),
    "spv_dec_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_int32, C.c_void_p]),
    "spv_dec_materialize": (C.c_int, [C.POINTER(SpvDecParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "spv_dec_dz": (C.c_int, [C.POINTER(SpvDecParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spv_dec_softmax_bwd": (C.c_int, [C.POINTER(SpvDecParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spv_dec_heads_bwd": (C.c_int, [C.POINTER(SpvDecParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spv_dec_tables_grouped": (C.c_int, [C.POINTER(SpvDecGroup), C.c_int32, C.c_void_p]),
    "spv_dec_logits_grouped": (C.c_int, [C.POINTER(SpvDecGroup), C.c_int32, C.c_void_p]),
    "spv_dec_lse_grouped": (C.c_int, [C.POINTER(SpvDecGroup), C.c_int32, C.c_void_p]),
    "spv_dec_nb_fwd_grouped": (C.c_int, [C.POINTER(SpvDecGroup), C.c_int32, C.c_int32, C.c_void_p]),
    "spv_dec_heads_bwd_grouped": (C.c_int, [C.POINTER(SpvDecGroup), C.c_int32, C.c_void_p]),
    "spv_gemm_bf16_grouped": (C.c_int, [C.POINTER(SpvGemmArgs), C.c_int32, C.c_void_p]),
    "spv_linear_fwd": (C.c_int, [C.POINTER(SpvLinearBatch), C.c_void_p]),
    "spv_linear_dgrad": (C.c_int, [C.POINTER(SpvLinearBatch), C.c_void_p]),
    "spv_linear_wgrad": (C.c_int, [C.POINTER(SpvLinearBatch), C.c_void_p, C.c_int64, C.c_void_p]),
    "spv_bn_fwd": (C.c_int, [C.POINTER(SpvBnBatch), C.c_void_p]),
    "spv_bn_bwd": (C.c_int, [C.POINTER(SpvBnBatch), C.c_void_p]),
    "spv_enc_sample_fwd": (C.c_int, [C.POINTER(SpvSampleBatch), C.c_void_p]),
    "spv_enc_sample_bwd": (C.c_int, [C.POINTER(SpvSampleBatch), C.c_void_p]),
    "spv_enc_heads_fwd": (C.c_int, [C.POINTER(SpvBnBatch), C.POINTER(SpvSampleBatch), C.c_void_p]),
    "spv_enc_heads_bwd": (C.c_int, [C.POINTER(SpvBnBatch), C.POINTER(SpvSampleBatch), C.c_void_p]),
    "spv_poe_partner": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spv_poe_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spv_poe_fuse_fwd": (C.c_int, [C.POINTER(SpvPoeArgs), C.c_void_p]),
    "spv_poe_fuse_bwd": (C.c_int, [C.POINTER(SpvPoeArgs), C.c_void_p]),
    "spv_plan_invmap": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "spv_plan_argmax": (C.c_int, [C.POINTER(SpvPlan), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "spv_plan_expert_fwd": (C.c_int, [C.POINTER(SpvPlanExpertArgs), C.c_void_p]),
    "spv_plan_expert_bwd": (C.c_int, [C.POINTER(SpvPlanExpertArgs), C.c_void_p]),
    "spv_poe_comp_fwd": (C.c_int, [C.POINTER(SpvPoeCompArgs), C.c_void_p]),
    "spv_poe_comp_bwd": (C.c_int, [C.POINTER(SpvPoeCompArgs), C.c_void_p]),
    "spv_zsplit_fwd": (C.c_int, [C.POINTER(SpvZsplitArgs), C.c_void_p]),
    "spv_zsplit_bwd": (C.c_int, [C.POINTER(SpvZsplitArgs), C.c_void_p]),
    "spv_bn_fold_fwd": (C.c_int, [C.POINTER(SpvFoldBatch), C.c_void_p]),
    "spv_bn_fold_bwd": (C.c_int, [C.POINTER(SpvFoldBatch), C.c_void_p]),
    "spv_trunk_fold_fwd": (C.c_int, [C.POINTER(SpvTrunkBatch), C.c_void_p]),
    "spv_trunk_fold_bwd": (C.c_int, [C.POINTER(SpvTrunkBatch), C.c_void_p]),
    "spv_reduce_slabs": (C.c_int, [C.POINTER(SpvReduceBatch), C.c_void_p]),
    "spv_loss_assemble": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "spv_adam_step_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                       C.c_void_p]),
    "spv_gather_u32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "spv_randn": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64, C.c_void_p]),
    "spv_counter_bump": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spv_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


class SpvError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library.  Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpvError(
            f"{LIB_PATH} is missing: build it with `python -m spvipes_amd.build` (hipcc, gfx950). "
            "spvipes_amd has no CPU or eager-PyTorch fallback for the hot path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export the symbol
        fn.restype, fn.argtypes = res, args
    # a library built from other sources or with other compiler flags than the ones next to it never loads silently
    from .build import build_id

    have, want = (lib.spv_build_id() or b"").decode(), build_id()
    if have != want and os.environ.get("SPV_ALLOW_STALE_LIB", "0") != "1":
        raise SpvError(f"{LIB_PATH} is stale (built from fingerprint {have}, the sources and flags now give {want}): "
                       "rebuild with `python -m spvipes_amd.build`")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().spv_last_error()
        raise SpvError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


# ---- optional per-entry-point timing with HIP events on the launch stream (bench.py roofline) ----
_PROFILE = None  # None = off; else {entry point name: [(start_event, end_event), ...]}


def profile_start(names) -> None:
    global _PROFILE
    _PROFILE = {n: [] for n in names}


def profile_stop() -> dict:
    """Returns {name: [milliseconds per call]} and switches profiling off (synchronises)."""
    global _PROFILE
    prof, _PROFILE = _PROFILE or {}, None
    torch.cuda.synchronize()
    return {n: [s.elapsed_time(e) for s, e in evs] for n, evs in prof.items()}


def call(name: str, *args) -> None:
    """Invoke one C-ABI entry point on torch's current stream; raise on a non-zero status."""
    fn = getattr(load(), name)
    if _PROFILE is not None and name in _PROFILE:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        _PROFILE[name].append((s, e))
    else:
        rc = fn(*args)
    check(rc, name)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise SpvError("spvipes_amd kernels need tensors resident in HBM (got a CPU tensor)")
    return t.data_ptr()


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m
