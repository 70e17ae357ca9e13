"""The C-ABI of libroreg_hip.so as data: ctypes prototypes (name -> (restype, argtypes), mirroring include/roreg_hip.h declaration by
declaration -- tests/test_host_logic.py checks the two against each other and against the exported symbols) and the numpy layouts of the task
structs the batched entry points take.  roreg_amd/hip.py binds them."""
import ctypes
from ctypes import c_int, c_void_p, c_size_t, c_double, c_float, c_char_p

import numpy as np

ABI_VERSION = 6          # == ROREG_ABI_VERSION of include/roreg_hip.h; hip.lib() refuses a library that reports another one

_P = c_void_p
PROTOTYPES = {
    'roreg_abi_version': (c_int, []),
    'roreg_last_error': (c_char_p, []),
    'roreg_set_group_tables': (c_int, [_P, _P, _P]),
    'roreg_group_conv_packed_size': (c_size_t, [c_int, c_int, c_int]),
    'roreg_group_conv_pack_weights': (c_int, [_P, c_int, c_int, c_int, _P]),
    'roreg_group_conv_workspace_size': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'roreg_group_conv': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    'roreg_group_conv_split': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'roreg_group_conv_f16x2': (c_int, [_P, _P, c_int, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'roreg_group_conv_f16x2_packed': (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'roreg_dense_f16x2': (c_int, [_P, _P, c_int, _P, _P, _P, c_float, c_float, _P, _P, c_int, _P, _P, c_int, c_int, c_int, _P]),
    'roreg_dense_split': (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P]),
    'roreg_gf_finalize': (c_int, [_P, _P, c_int, _P, c_int, _P]),
    'roreg_det_score': (c_int, [_P, _P, c_int, _P]),
    'roreg_inv_descriptor': (c_int, [_P, c_int, _P, c_int, _P]),
    'roreg_nn_search': (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    'roreg_nn_search_ex': (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    'roreg_pdist': (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, _P]),
    'roreg_knn_search_workspace': (c_size_t, [c_int, c_int]),
    'roreg_knn_search': (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    'roreg_knn_search_ex': (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    'roreg_knn_search_seg_workspace': (c_size_t, [ctypes.c_longlong, c_int, c_int, c_int]),
    'roreg_knn_search_seg': (c_int, [_P, _P, _P, _P, c_int, ctypes.c_longlong, c_int, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    'roreg_mutual_matches': (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    'roreg_mutual_match_batch_workspace': (c_size_t, [c_int, c_int]),
    'roreg_mutual_match_batch': (c_int, [_P, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    'roreg_des2r': (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P]),
    'roreg_set_des2r_tables': (c_int, [c_int, _P, _P, _P, _P]),
    'roreg_group_corr_irrep': (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P]),
    'roreg_group_corr_mfma': (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P]),
    'roreg_des2r_irrep': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    'roreg_des2r_recheck_count': (c_int, [c_int, _P]),
    'roreg_feat_coefs': (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    'roreg_et_gather': (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, c_int, _P, _P]),
    'roreg_quat_to_trans': (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    'roreg_ransac_score': (c_int, [_P, _P, _P, c_int, c_int, _P, _P, c_int, c_double, _P, _P, _P, _P]),
    'roreg_refine': (c_int, [_P, _P, _P, c_int, c_int, _P, c_int, _P, _P, _P, c_double, _P, _P, _P]),
    'roreg_lt_prepare_batch': (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'roreg_lt_finish_batch': (c_int, [_P, c_int, c_int, _P, _P, _P, _P]),
    'roreg_ransac_batch_workspace': (c_size_t, [c_int, ctypes.c_longlong, c_int]),
    'roreg_ransac_batch': (c_int, [_P, c_int, ctypes.c_longlong, c_int, c_int, c_double, c_int, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    'roreg_refine_batch': (c_int, [_P, _P, c_int, ctypes.c_longlong, _P, ctypes.c_double, c_int, _P, _P, _P, _P]),
    'roreg_yohoc_draw': (c_int, [_P, ctypes.c_longlong, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    'roreg_gather_rows_batch': (c_int, [_P, c_int, c_int, c_int, _P]),
    'roreg_gather_rows_f64': (c_int, [_P, _P, c_int, c_int, _P, _P]),
    'roreg_group_corr': (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    'roreg_topk_dot_workspace_size': (c_size_t, [c_int, c_int, c_int]),
    'roreg_topk_dot': (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, _P, c_size_t, _P, _P, c_int, c_int, c_int, _P]),
    'roreg_context_colmax': (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, _P]),
    'roreg_sinkhorn_batch_workspace_size': (c_size_t, [c_int, c_int, c_int, ctypes.c_longlong, ctypes.c_longlong]),
    'roreg_sinkhorn_batch_consts': (c_int, [_P, _P, c_int, _P]),
    'roreg_sinkhorn_batch': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    'roreg_sinkhorn_batch2_workspace_size': (c_size_t, [c_int, c_int, c_int, ctypes.c_longlong, ctypes.c_longlong]),
    'roreg_sinkhorn_batch2': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_int, _P, _P, _P, _P, _P, c_size_t, c_int, _P]),
    'roreg_sinkhorn_batch3_workspace_size': (c_size_t, [c_int, c_int, c_int, ctypes.c_longlong, ctypes.c_longlong, c_int, c_int]),
    'roreg_sinkhorn_batch3': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_int, _P, _P, _P, _P, _P, c_size_t, c_int, _P, _P]),
    'roreg_write_files': (c_int, [_P, _P, _P, _P, _P, c_int, c_int]),
    'roreg_yohoc_draw_many': (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, _P, _P, c_int]),
    'roreg_mlp_head_workspace': (c_size_t, [c_int, c_int, c_int]),
    'roreg_mlp_head': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P, _P, _P]),
    'roreg_sinkhorn_early_exit': (c_int, [c_int]),
    'roreg_sinkhorn_iteration_stats': (c_int, [_P, _P, c_int, _P]),
    'roreg_linear': (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, _P]),
    'roreg_linear_mfma': (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, _P]),
    'roreg_linear_path': (c_int, [c_int]),
    'roreg_linear_cat3': (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, c_int, _P, _P]),
    'roreg_gemm_persistent': (c_int, [c_int]),
    'roreg_instnorm_stats': (c_int, [_P, c_int, c_int, c_float, _P, _P, _P, c_int, c_int, _P]),
    'roreg_mlp_tail': (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    'roreg_mlp_tail_mfma': (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    'roreg_knn_attention': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    'roreg_rm_elementwise': (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    'roreg_sinkhorn_workspace_size': (c_size_t, [c_int, c_int]),
    'roreg_sinkhorn': (c_int, [_P, c_int, _P, c_int, c_float, c_int, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    'roreg_profile_enable': (c_int, [c_int]),
    'roreg_profile_read': (c_int, [c_int, _P, _P]),
    'roreg_set_fourier_tables': (c_int, [_P]),
    'roreg_mt_shuffle_prefix': (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int]),
    'roreg_mt_stream_shuffle_prefix': (c_int, [_P, _P, _P, c_int, c_int, _P]),
    'roreg_irrep_gemm_tiles': (c_size_t, [c_int, c_int, _P]),
    'roreg_irrep_gemm_tiles_m': (c_size_t, [c_int, c_int, c_int, _P]),
    'roreg_irrep_gemm': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    'roreg_irrep_gemm_split': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    'roreg_irrep_gemm_f16x2': (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P]),
    'roreg_row_bound': (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, _P]),
    'roreg_ft_nonlin': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P]),
    'roreg_ft_nonlin_packed': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
}


_RANSAC_TASK = np.dtype([('keys0', np.uint64), ('keys1', np.uint64), ('matches', np.uint64), ('w', np.uint64), ('Trans', np.uint64),
                         ('hyp_rows', np.uint64), ('M', np.int32), ('H', np.int32), ('koff', np.int64)])

_MATCH_TASK = np.dtype([('desc0', np.uint64), ('desc1', np.uint64), ('rows0', np.uint64), ('rows1', np.uint64), ('m0', np.int32), ('m1', np.int32)])

_GATHER_TASK = np.dtype([('src', np.uint64), ('rows', np.uint64), ('dst', np.uint64), ('n', np.int32), ('pad', np.int32)])

_LT_TASK = np.dtype([('before0', np.uint64), ('before1', np.uint64), ('after0', np.uint64), ('after1', np.uint64), ('keys0', np.uint64),
                     ('keys1', np.uint64), ('matches', np.uint64), ('sel', np.uint64), ('n', np.int32), ('pad', np.int32), ('off', np.int64),
                     ('coef0', np.uint64), ('coef1', np.uint64)])
