"""ctypes binding of libctrhip.so (the C ABI declared in include/ctrhip.h).

The library is built in-tree by `__graft_entry__.build()` (or `make -C recommendation_amd/csrc`).
There is no fallback: if the shared object is missing the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTRHIP_LIB: another build of the same library (A/B timing of a kernel change); default: the in-tree build
LIB_PATH = os.environ.get('CTRHIP_LIB') or os.path.join(_HERE, 'lib', 'libctrhip.so')

c_f32p = C.c_void_p   # device pointers travel as integers
c_i32p = C.c_void_p
c_i64p = C.c_void_p
c_u32p = C.c_void_p
vp = C.c_void_p


class Dropout(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('stream_id', C.c_uint64),
                ('keep_threshold', C.c_uint32), ('scale', C.c_float), ('seed_add', C.c_void_p)]


class Opt(C.Structure):
    _fields_ = [('kind', C.c_int), ('lr', C.c_float), ('beta1', C.c_float),
                ('beta2', C.c_float), ('eps', C.c_float), ('l1', C.c_float), ('l2', C.c_float),
                ('lr_dev', C.c_void_p)]


class FmBwd(C.Structure):
    _fields_ = [('g_first', C.c_void_p), ('g_first_ld', C.c_int64), ('g_second', C.c_void_p),
                ('g_second_ld', C.c_int64), ('s', C.c_void_p), ('z', C.c_void_p), ('z_ld', C.c_int64),
                ('n_rows', C.c_int), ('E', C.c_int)]


class Epilogue(C.Structure):
    _fields_ = [('bias', C.c_void_p), ('relu', C.c_int), ('accumulate', C.c_int),
                ('dropout', C.POINTER(Dropout)), ('mask_src', C.c_void_p),
                ('mask_ld', C.c_int64), ('mask_scale', C.c_float), ('fm_bwd', C.POINTER(FmBwd)),
                ('a_amax', C.c_void_p), ('b_amax', C.c_void_p), ('a_amax_floor', C.c_float),
                ('b_amax_floor', C.c_float), ('c_amax_out', C.c_void_p)]


OPT_GRAD_ONLY, OPT_ADAM, OPT_ADAGRAD, OPT_SGD, OPT_FTRL = 0, 1, 2, 3, 4
COMM_ID_BYTES = 128          # CTR_COMM_ID_BYTES
LOSS_SIGMOID_LOGLOSS, LOSS_IDENTITY_LOGLOSS, LOSS_SIGMOID_SCE_PROB, LOSS_SCE_LOGIT, LOSS_PROB_SCE = 0, 1, 2, 3, 4

i, i64, sz, f = C.c_int, C.c_int64, C.c_size_t, C.c_float
DP = C.POINTER(Dropout)

# name -> (restype, argtypes); mirrors include/ctrhip.h one to one
SIGNATURES = {
    'ctr_last_error': (C.c_char_p, []),
    'ctr_version': (i, []),
    'ctr_keep_hash24': (C.c_uint32, [C.c_uint64, C.c_uint64, C.c_uint64]),
    'ctr_embed_fwd': (i, [vp, vp, vp, vp, vp, i, i, i, i, vp, i64, i, vp, i64, vp, vp, i64, DP, DP, vp, vp]),
    'ctr_embed_fwd_amax': (i, [vp, vp, vp, vp, vp, i, i, i, i, vp, i64, i, vp, i64, vp, vp, i64, DP, DP, vp, vp, vp]),
    'ctr_fm_fwd': (i, [vp, i64, i, i, i, vp, i64, vp, vp, i64, DP, DP, vp]),
    'ctr_relu_mask_apply': (i, [vp, i64, vp, i64, i, i, f, vp]),
    'ctr_embed_onehot_fwd': (i, [vp, vp, vp, vp, i, i, i, i, vp, i64, i, vp, vp]),
    'ctr_embed_bwd_dz': (i, [vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i, i, i, DP, DP, vp, i64, vp]),
    'ctr_embed_bwd_dz_blocks': (i, [vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i, i, i, DP, DP, vp, i64, vp, vp, vp, i, vp]),
    'ctr_embed_assemble_fwd': (i, [vp, vp, vp, vp, vp, i, i, i, i, vp, i64, i, vp, i64, vp, vp, i64, DP, DP, vp]),
    'ctr_comm_unique_id': (i, [vp]),
    'ctr_comm_init': (i, [vp, i, i, C.POINTER(vp)]),
    'ctr_comm_destroy': (i, [vp]),
    'ctr_comm_world': (i, [vp]),
    'ctr_comm_rank': (i, [vp]),
    'ctr_allreduce_sum_f32': (i, [vp, vp, i64, vp]),
    'ctr_all2all': (i, [vp, vp, vp, i64, vp]),
    'ctr_all2allv': (i, [vp, vp, vp, vp, vp, vp, vp, vp]),
    'ctr_criteo_tsv_count': (i64, [vp, i64]),
    'ctr_criteo_tsv_parse': (i, [vp, i64, vp, i, i, i64, vp, vp, vp, vp]),
    'ctr_emb_sort_workspace_bytes': (sz, [i64]),
    'ctr_emb_sort_workspace_bytes_fields': (sz, [i, i, i64, i]),
    'ctr_emb_sort': (i, [vp, vp, i, i, i, i64, i, vp, vp, vp, sz, vp]),
    'ctr_emb_bag_sort': (i, [vp, vp, i64, vp, i, i, i, i64, vp, vp, vp, sz, vp]),
    'ctr_embed_bag_fwd': (i, [vp, vp, vp, vp, i64, vp, vp, i, i, i, i, vp, i64, i, vp, i64, vp, vp, i64, DP, DP, vp, vp]),
    'ctr_emb_sort_static': (i, [vp, vp, i, i, i, i64, i, vp, vp, vp]),
    'ctr_sort_pairs': (i, [vp, vp, i64, i, vp, vp, vp, sz, vp]),
    'ctr_emb_sparse_apply_workspace_bytes': (sz, [i64, i]),
    'ctr_emb_sparse_apply': (i, [vp, vp, i64, i, i, i64, vp, i64, vp, vp, vp, vp, C.POINTER(Opt), vp, sz, vp]),
    'ctr_emb_sparse_apply_ld': (i, [vp, vp, i64, i, i, i64, vp, i64, vp, vp, vp, vp, i64, C.POINTER(Opt), vp, sz, vp]),
    'ctr_mark_rows': (i, [vp, i64, i64, vp, i, vp]),
    'ctr_adam_untouched_rows': (i, [vp, vp, vp, i64, i, vp, C.POINTER(Opt), vp]),
    'ctr_emb_sparse_apply_fm_workspace_bytes': (sz, [i64, i]),
    'ctr_emb_sparse_apply_fm': (i, [vp, vp, i64, i, i, i64, vp, i64, vp, i64, vp, vp, vp, i64, C.POINTER(Opt), vp, sz, vp]),
    'ctr_gemm_f32_workspace_bytes': (sz, [i, i, i]),
    'ctr_gemm_f32': (i, [i, i, i, i, i, vp, i64, vp, i64, vp, i64, C.POINTER(Epilogue), vp, sz, vp]),
    'ctr_dense_wgrad_workspace_bytes': (sz, [i, i, i]),
    'ctr_dense_wgrad': (i, [i, i, i, vp, i64, vp, i64, vp, vp, vp, sz, vp]),
    'ctr_dense_wgrad_scaled': (i, [i, i, i, vp, i64, vp, i64, vp, vp, vp, f, vp, vp, sz, vp]),
    'ctr_set_gemm_f16_split': (i, [i]),
    'ctr_absmax': (i, [vp, i64, i64, i, vp, vp]),
    'ctr_absmax_tensors': (i, [vp, vp, vp, i, vp, vp]),
    'ctr_colsum_workspace_bytes': (sz, [i, i]),
    'ctr_colsum': (i, [vp, i64, vp, i64, i, i, i, vp, i, vp, sz, vp]),
    'ctr_rowdot_fwd': (i, [vp, i64, vp, vp, i, i, vp, vp]),
    'ctr_rowdot_bwd_x': (i, [vp, vp, i, i, vp, i64, i, vp]),
    'ctr_loss_head_workspace_bytes': (sz, [i]),
    'ctr_loss_head': (i, [i, vp, vp, i, f, vp, vp, vp, vp, sz, vp]),
    'ctr_esmm_head': (i, [vp, vp, vp, i, f, vp, vp, vp, vp, vp, sz, vp]),
    'ctr_dense_update': (i, [vp, vp, vp, vp, i64, C.POINTER(Opt), vp]),
    'ctr_set_step_words': (i, [vp, C.c_uint64, vp, C.c_float, vp]),
    'ctr_cross_fwd_workspace_bytes': (sz, [i, i]),
    'ctr_cross_fwd': (i, [vp, i64, vp, vp, i64, i, i, i, vp, i64, vp, vp, sz, vp]),
    'ctr_cross_bwd_workspace_bytes': (sz, [i, i, i]),
    'ctr_cross_bwd': (i, [vp, i64, vp, vp, i64, vp, vp, i64, i, i, i, vp, i64, i, vp, vp, vp, sz, vp]),
    'ctr_pnn_inner_fwd': (i, [vp, i64, vp, i, i, i, i, vp, vp]),
    'ctr_pnn_inner_bwd_workspace_bytes': (sz, [i, i, i]),
    'ctr_pnn_inner_bwd': (i, [vp, i64, vp, vp, i64, i, i, i, i, vp, i64, vp, vp, sz, vp]),
    'ctr_outer_fwd': (i, [vp, i, i, vp, vp]),
    'ctr_outer_bwd': (i, [vp, vp, i, i, vp, vp]),
    'ctr_bias_act': (i, [vp, i64, i, i, vp, vp, i, DP, vp]),
    'ctr_vec_op': (i, [i, vp, vp, vp, i, vp]),
    'ctr_afm_pair_fwd': (i, [vp, i64, i, i, i, DP, vp, vp]),
    'ctr_afm_pair_bwd': (i, [vp, i64, vp, i, i, i, DP, vp, i64, vp]),
    'ctr_set_gemm_mode': (i, [i]),
    'ctr_get_gemm_mode': (i, []),
    'ctr_gemm_split_blocks_per_cu': (i, []),
    'ctr_unit_bwd_supported': (i, [i]),
    'ctr_unit_bwd_workspace_bytes': (sz, [i, i]),
    'ctr_unit_bwd': (i, [i, i, vp, vp, vp, i64, vp, i64, i, f, vp, vp, vp, sz, vp]),
    'ctr_unit_bwd_keep': (i, [i, i, vp, vp, vp, i64, vp, i64, i, f, i, DP, i, DP, vp, vp, vp, vp, sz, vp]),
    'ctr_mlp_chain_supported': (i, [i, i, i, i]),
    'ctr_mlp_chain_fwd': (i, [i, i, vp, i64, i, vp, vp, DP, vp, i64, i, vp, vp, DP, vp, i64, i, vp, vp, DP, vp, i64, vp]),
    'ctr_mlp_chain_bwd_supported': (i, [i, i, i]),
    'ctr_tail_wgrad2_supported': (i, [i, i, i, i]),
    'ctr_tail_bwd_fused_supported': (i, [i, i, i]),
    'ctr_tail_bwd_fused_workspace_bytes': (sz, [i, i, i, i]),
    'ctr_tail_bwd_fused': (i, [i, i, vp, i64, i, vp, vp, i64, f, i, vp, vp, i64, f, vp, i64, vp, vp, vp, vp, vp, sz, vp]),
    'ctr_tail_wgrad2_workspace_bytes': (sz, [i, i, i, i, i]),
    'ctr_tail_wgrad2': (i, [i, i, vp, i64, i, vp, i64, vp, vp, i, vp, i64, i, vp, i64, vp, vp, vp, sz, vp]),
    'ctr_mlp_chain_bwd': (i, [i, i, vp, i64, i, vp, vp, i64, f, vp, i64, i, vp, vp, i64, f, vp, i64, vp, vp]),
    'ctr_auc_workspace_bytes': (sz, [i64]),
    'ctr_auc': (i, [vp, vp, i64, vp, vp, vp, sz, vp]),
    'ctr_shard_route': (i, [vp, vp, vp, i, i, i, vp, vp, vp, vp]),
    'ctr_shard_bucket': (i, [vp, vp, vp, i64, i, i64, vp, vp, vp, vp, i64, i, i, vp]),
    'ctr_rows_gather': (i, [vp, i64, vp, i64, i, i, vp, vp]),
    'ctr_rows_scatter': (i, [vp, vp, i64, i, i, vp, i64, vp]),
    'ctr_afm_attn_fwd': (i, [vp, i64, i, i, i, i, vp, vp, vp, DP, vp, vp]),
    'ctr_afm_attn_bwd_workspace_bytes': (sz, [i, i, i]),
    'ctr_afm_attn_bwd': (i, [vp, i64, vp, i, i, i, i, vp, vp, vp, DP, vp, i64, vp, vp, vp, vp, sz, vp]),
    'ctr_din_build_idx': (i, [vp, vp, i64, i, vp, vp]),
    'ctr_din_concat_fwd': (i, [vp, vp, i, i, i, vp, vp]),
    'ctr_din_concat_bwd': (i, [vp, vp, vp, i, i, i, vp, i, vp, i, vp]),
    'ctr_din_attn_pool_fwd': (i, [vp, vp, vp, i, i, i, vp, vp, vp]),
    'ctr_din_attn_mlp_supported': (i, [i, i, i]),
    'ctr_din_attn_mlp_fwd': (i, [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'ctr_din_attn_mlp_bwd': (i, [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, vp]),
    'ctr_din_attn_mlp_wgrad_workspace_bytes': (sz, [i, i]),
    'ctr_din_attn_mlp_wgrad': (i, [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'ctr_din_attn_mlp_fwd_sl': (i, [vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'ctr_din_attn_mlp_bwd_sl': (i, [vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, vp]),
    'ctr_din_attn_mlp_wgrad_sl': (i, [vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'ctr_din_attn_pool_bwd': (i, [vp, vp, vp, vp, i, i, i, vp, vp, i, vp]),
    'ctr_bn_infer': (i, [vp, i64, i, i, vp, vp, vp, vp, f, vp, i64, vp, i64, vp]),
    'ctr_dice_fwd': (i, [vp, i64, i, i, vp, f, vp, i64, vp]),
    'ctr_dice_bwd': (i, [vp, i64, i, i, vp, f, vp, i64, vp, i64, vp, i64, vp]),
    'ctr_bn_infer_bwd': (i, [vp, i64, vp, i64, i, i, vp, vp, f, vp, i64, vp, vp, vp]),
    'ctr_dice_bwd_sum': (i, [vp, i64, i, i, vp, f, vp, i64, vp, i64, vp, vp]),
    'ctr_dsigmoid_apply': (i, [vp, i64, vp, i64, i, i, vp]),
    'ctr_mean_positive_diff': (i, [vp, vp, i, vp, vp]),
    'ctr_copy2d': (i, [vp, i64, vp, i64, i, i, i, vp]),
    'ctr_mul2d': (i, [vp, i64, vp, i64, vp, i64, i, i, vp]),
    'ctr_dropout_apply': (i, [vp, i64, i, i, DP, vp]),
    'ctr_dropout_apply2': (i, [vp, i64, i, i, DP, i, DP, vp]),
}

_lib = None


def load():
    """Load libctrhip.so and attach the argument types; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libctrhip.so is not built (%s missing). Build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or `make -C recommendation_amd/csrc`. '
            'There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class CtrError(RuntimeError):
    pass


def check(rc, what=''):
    if rc != 0:
        msg = load().ctr_last_error().decode('utf-8', 'replace')
        raise CtrError('%s failed with code %d: %s' % (what or 'libctrhip call', rc, msg))
