"""ctypes binding of include/elliot_hip.h.

There is no CPU fallback: if libelliot_hip.so is missing or fails to load, every product
entry point raises.  (oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EL_LIB_PATH") or os.path.join(_HERE, "csrc", "libelliot_hip.so")   # (EL_LIB_PATH: another BUILD of the
#                                                       same library for kernel A/B runs, scripts/exp/build_variants.sh -- never a fallback)

EL_OPT_ADAM_TF_DENSE = 0
EL_OPT_ADAM_LAZY = 2
EL_OPT_SGD = 3
EL_TOPK_AUTO = 0
EL_TOPK_MFMA = 1
EL_TOPK_SIMPLE = 2
EL_TOPK_SCREEN = 3
EL_BPR_AUTO = 0
EL_BPR_ATOMIC = 1
EL_BPR_SORTED = 2

_f32p = C.c_void_p
_i32p = C.c_void_p
_i64p = C.c_void_p
_f64p = C.c_void_p


class BprmfState(C.Structure):
    _fields_ = [
        ("Gu", _f32p), ("Gi", _f32p), ("Bi", _f32p),
        ("gGu", _f32p), ("gGi", _f32p), ("gBi", _f32p),
        ("mGu", _f32p), ("vGu", _f32p), ("mGi", _f32p), ("vGi", _f32p), ("mBi", _f32p), ("vBi", _f32p),
        ("tGu", _i32p), ("tGi", _i32p), ("tBi", _i32p),
        ("U", C.c_int64), ("I", C.c_int64), ("F", C.c_int32),
        ("uslot", _i64p), ("gGu_rows", _f32p), ("gGu_cap", C.c_int64), ("Gu_next", _f32p),
        ("Gu_last", _i32p), ("Gu_old", _f32p), ("Gu_old_cap", C.c_int64), ("lr_hist", _f32p), ("lr_hist_cap", C.c_int32),
        ("Gi_last", _i32p), ("Gi_defer", C.c_int32), ("replay_series", C.c_int32),
    ]


class BprsgdState(C.Structure):
    _fields_ = [
        ("P", _f64p), ("Q", _f64p), ("b", _f64p),
        ("U", C.c_int64), ("I", C.c_int64), ("F", C.c_int32),
        ("lr", C.c_double), ("reg_bias", C.c_double), ("reg_user", C.c_double),
        ("reg_pos", C.c_double), ("reg_neg", C.c_double),
    ]


class VaeState(C.Structure):
    _fields_ = [
        ("I", C.c_int64), ("H", C.c_int32), ("L", C.c_int32), ("Bmax", C.c_int64),
        ("w", C.c_void_p * 8), ("g", C.c_void_p * 8), ("m", C.c_void_p * 8), ("v", C.c_void_p * 8),
        ("h", _f32p), ("mv", _f32p), ("z", _f32p), ("dz", _f32p), ("h2", _f32p), ("logits", _f32p),
        ("dh2", _f32p), ("dmv", _f32p), ("dh", _f32p), ("rnorm", _f32p),
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("dae", C.c_int32),
    ]


_P4 = C.c_void_p * 4


class GraphCsr(C.Structure):
    _fields_ = [
        ("indptr", _i64p), ("indices", _i32p), ("vals", _f32p), ("N", C.c_int64), ("n0", C.c_int64),
        ("chunk_row", _i32p), ("chunk_lo", _i64p), ("chunk_slot", _i32p), ("n_chunks", C.c_int64),
        ("multi_row", _i32p), ("multi_slot", _i32p), ("multi_cnt", _i32p), ("n_multi", C.c_int64), ("part", _f32p),
    ]


class Mf2020State(C.Structure):
    _fields_ = [
        ("P", _f64p), ("Q", _f64p), ("bu", _f64p), ("bi", _f64p), ("gb", _f64p),
        ("U", C.c_int64), ("I", C.c_int64), ("F", C.c_int32), ("lr", C.c_double), ("reg", C.c_double),
    ]


class NmfState(C.Structure):
    _fields_ = [
        ("U", C.c_int64), ("I", C.c_int64), ("Bmax", C.c_int64),
        ("F", C.c_int32), ("E", C.c_int32), ("n_layers", C.c_int32), ("use_mf", C.c_int32), ("use_mlp", C.c_int32),
        ("head_bias", C.c_int32), ("units", C.c_int32 * 4),
        ("tab", _P4), ("gtab", _P4), ("mtab", _P4), ("vtab", _P4),
        ("W", _P4), ("b", _P4), ("gW", _P4), ("gb", _P4), ("mW", _P4), ("vW", _P4), ("mb", _P4), ("vb", _P4),
        ("hw", _f32p), ("hb", _f32p), ("ghw", _f32p), ("ghb", _f32p), ("mhw", _f32p), ("vhw", _f32p),
        ("mhb", _f32p), ("vhb", _f32p),
        ("X0", _f32p), ("dX0", _f32p), ("MF", _f32p), ("dlogit", _f32p),
        ("act", _P4), ("dact", _P4),
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("dropout", C.c_float), ("drop_step", C.c_int32), ("drop_seed", C.c_uint64),
        ("row_last", C.c_void_p * 2), ("row_stamp", C.c_void_p * 2), ("row_own", C.c_void_p), ("lr_hist", C.c_void_p),
        ("lr_hist_cap", C.c_int32), ("hist_base", C.c_int32), ("opt_step", C.c_int32), ("flushed_step", C.c_int32),
        ("claim_seq", C.c_int32),
        ("batch_u", C.c_void_p), ("batch_i", C.c_void_p), ("batch_n", C.c_int64),
        ("step_ws", C.c_void_p), ("step_ws_bytes", C.c_size_t),
        ("pre_u", C.c_void_p), ("pre_i", C.c_void_p), ("pre_n", C.c_int64), ("sort_set", C.c_int32),
        ("replay_series", C.c_int32),
    ]


class PwmfState(C.Structure):
    _fields_ = [
        ("U", C.c_int64), ("I", C.c_int64), ("F", C.c_int32), ("kind", C.c_int32), ("alpha", C.c_float), ("l_w", C.c_float),
        ("Gu", _f32p), ("Gi", _f32p), ("Bu", _f32p), ("Bi", _f32p),
        ("gGu", _f32p), ("gGi", _f32p), ("gBu", _f32p), ("gBi", _f32p),
        ("mGu", _f32p), ("mGi", _f32p), ("mBu", _f32p), ("mBi", _f32p),
        ("vGu", _f32p), ("vGi", _f32p), ("vBu", _f32p), ("vBi", _f32p),
    ]


EL_TOPK_ITEMS_UNCHANGED = 0x100
EL_NMF_SCREEN = 0x200
EL_PW_MSE, EL_PW_MSE_SIGMOID, EL_PW_LOGISTIC = 0, 1, 2
EL_PW_ADAM, EL_PW_ADAGRAD = 0, 1
EL_PW_BOTH, EL_PW_ITEMS, EL_PW_USERS = 0, 1, 2

# name -> (restype, argtypes); mirrors include/elliot_hip.h one to one
PROTOTYPES = {
    "el_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "el_ctx_destroy": (C.c_int, [C.c_void_p]),
    "el_last_error": (C.c_char_p, []),
    "el_abi_version": (C.c_int, []),
    "el_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "el_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "el_timing_filter": (C.c_int, [C.c_void_p, C.c_char_p]),
    "el_tuning_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "el_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "el_ctx_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]),
    "el_comm_unique_id": (C.c_int, [C.c_void_p]),
    "el_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "el_comm_destroy": (C.c_int, [C.c_void_p]),
    "el_comm_rank": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "el_allreduce_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _f32p, C.c_int64]),
    "el_reduce_scatter_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_int64]),
    "el_allgather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "el_allgather_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _i32p, _f32p, C.c_int64, C.c_int32, _i32p, _f32p]),
    "el_host_split_flags": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_uint32, C.c_int32, C.c_void_p]),
    "el_host_split_flags_state": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_int32, C.c_void_p]),
    "el_host_negative_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "el_host_pyset_order": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]),
    "el_timing_report": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "el_bpr_sample": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, _i32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                C.c_uint64, C.c_uint64, C.c_int64, _i32p, _i32p, _i32p]),
    "el_bpr_sampler_meta_bytes": (C.c_size_t, [C.c_int64]),
    "el_bpr_sampler_meta_build": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, _i32p, C.c_int64, C.c_void_p]),
    "el_bpr_sample_meta": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, _i32p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                     C.c_uint64, C.c_uint64, C.c_int64, _i32p, _i32p, _i32p]),
    "el_bpr_sample_mt19937_ws_bytes": (C.c_size_t, [C.c_int64]),
    "el_bpr_sample_mt19937": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _i64p, _i32p, _i64p, _i32p, C.c_int64, C.c_int64,
                                        C.c_int64, _i32p, _i32p, _i32p, C.c_void_p, C.c_size_t]),
    "el_bprmf_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), _i32p, _i32p, _i32p, C.c_int64,
                                      C.c_float, C.c_float, C.c_float, C.c_int, C.c_int32, C.c_float, _f64p,
                                      C.c_int, C.c_void_p, C.c_size_t]),
    "el_bprmf_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    "el_bprmf_deterministic": (C.c_int, []),
    "el_bprmf_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), _i32p, _i32p, _i32p, C.c_int64,
                                 C.c_float, C.c_float, C.c_int32, _f64p, C.c_void_p, C.c_size_t]),
    "el_bprmf_grads_presorted": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), _i32p, _i32p, _i32p, C.c_int64,
                                 C.c_float, C.c_float, C.c_int32, _f64p, C.c_void_p, C.c_size_t]),
    "el_bprmf_train_step_presorted": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), _i32p, _i32p, _i32p, C.c_int64,
                                                C.c_float, C.c_float, C.c_float, C.c_int, C.c_int32, C.c_float, _f64p, C.c_void_p, C.c_size_t]),
    "el_bprmf_presort": (C.c_int, [C.c_void_p, C.c_void_p, _i32p, _i32p, _i32p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t]),
    "el_bprmf_shard_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), _i32p, _i32p, _i32p, C.c_int64,
                                       C.c_float, C.c_float, C.c_int32, _f32p, _f64p, C.c_void_p, C.c_size_t]),
    "el_rows_segment_sum_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "el_rows_segment_sum": (C.c_int, [C.c_void_p, C.c_void_p, _i32p, _f32p, C.c_int64, C.c_int32, C.c_int64, _f32p,
                                      C.c_void_p, C.c_size_t]),
    "el_bprmf_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), C.c_float, C.c_int, C.c_int32, C.c_float]),
    "el_spmm_csr_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(GraphCsr), _f32p, _f32p, C.c_int32, _f32p, _f32p]),
    "el_lightgcn_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "el_lightgcn_propagate": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(GraphCsr), _f32p, _f32p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]),
    "el_ngcf_pre": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_int64, C.c_int32, _f32p]),
    "el_ngcf_post": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_uint32, _f32p, _f32p,
                               _f32p, C.c_int32, C.c_int32]),
    "el_adam_l2_dense": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, _f32p, _f32p, C.c_int64, C.c_float, C.c_float]),
    "el_mf2020_train": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Mf2020State), _i32p, C.c_int64, _f64p]),
    "el_bprsgd_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprsgdState), _i32p, _i32p, _i32p,
                                  C.c_int64, C.c_int64]),
    "el_bprsgd_apply_levels": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprsgdState), _i32p, _i32p, _i32p,
                                         C.c_void_p, C.c_int64]),
    "el_bprsgd_levels_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "el_rec_metrics_ws_bytes": (C.c_size_t, [C.c_int64]),
    "el_topk_fragile": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_int32, C.c_int64, C.c_int64, _i32p, _f32p, C.c_int64,
                                  C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "el_rec_metrics": (C.c_int, [C.c_void_p, C.c_void_p, _i32p, C.c_int64, C.c_int64, C.c_int64, _i64p, _i32p, _f32p,
                                 C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "el_score_topk_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int]),
    "el_topk_screen_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "el_score_topk": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, _f32p, _f32p, C.c_int64, C.c_int64, C.c_int64,
                                C.c_int64, C.c_int32, _i64p, _i32p, _i64p, _i32p, C.c_int32, _i32p, _f32p,
                                C.c_int, C.c_void_p, C.c_size_t]),
    "el_score_topk_f64": (C.c_int, [C.c_void_p, C.c_void_p, _f64p, _f64p, _f64p, C.c_int64, C.c_int64, C.c_int64,
                                    C.c_int64, C.c_int32, _i64p, _i32p, _i64p, _i32p, C.c_int32, _i32p, _f64p]),
    "el_topk_merge": (C.c_int, [C.c_void_p, C.c_void_p, _i32p, _f32p, C.c_int32, C.c_int64, C.c_int32, _i32p, _f32p]),
    "el_gemm_ws_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "el_gemm_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, _f32p, C.c_int64,
                              _f32p, C.c_int64, _f32p, C.c_int64, _f32p, C.c_int, C.c_void_p, C.c_size_t]),
    "el_vae_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(VaeState), _i64p, _i32p, _i32p, C.c_int64, _f32p,
                                    C.c_float, C.c_float, C.c_uint64, C.c_int32, C.c_float, _f64p]),
    "el_vae_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(VaeState), _i64p, _i32p, _i32p, C.c_int64, C.c_int64, _f32p,
                               C.c_float, C.c_float, C.c_uint64, C.c_int32, _f64p]),
    "el_vae_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(VaeState), C.c_float]),
    "el_vae_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(VaeState), _i64p, _i32p, _i32p, C.c_int64, _f32p]),
    "el_pointwise_sample_meta": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, _i32p, C.c_void_p, C.c_int64, C.c_int64, C.c_uint64,
                                           C.c_uint64, C.c_int64, _i32p, _i32p, _f32p]),
    "el_pointwise_sample": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, _i32p, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64,
                                      C.c_int64, _i32p, _i32p, _f32p]),
    "el_nmf_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(NmfState), _i32p, _i32p, C.c_int64, _f32p]),
    "el_nmf_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(NmfState), _i32p, _i32p, _f32p, C.c_int64, C.c_int64, _f64p]),
    "el_nmf_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(NmfState), C.c_int32, C.c_float]),
    "el_nmf_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(NmfState), _i32p, _i32p, _f32p, C.c_int64,
                                    C.c_int32, C.c_float, _f64p]),
    "el_nmf_sync_tables": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(NmfState)]),
    "el_nmf_step_ws_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(NmfState)]),
    "el_nmf_presort": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(NmfState), C.c_void_p, C.c_void_p, C.c_int64]),
    "el_nmf_score_supported": (C.c_int, [C.POINTER(NmfState), C.c_int32]),
    "el_nmf_screen_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "el_nmf_score_ws_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(NmfState), C.c_int64, C.c_int64, C.c_int32, C.c_int]),
    "el_nmf_score_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(NmfState), C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                    _i64p, _i32p, _i64p, _i32p, C.c_int32, _i32p, _f32p, C.c_int, C.c_void_p, C.c_size_t]),
    "el_gmf_item_image": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_int64, C.c_int32, _f32p]),
    "el_bprmf_sync_users": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), C.c_int32]),
    "el_bprmf_sync_items": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), C.c_int32]),
    "el_selftest_replay_math": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "el_bprmf_train_loop_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "el_bprmf_train_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), _i64p, _i32p, C.c_uint64, C.c_uint64,
                                      C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int32, C.c_void_p,
                                      _f64p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "el_topk_rerank": (C.c_int, [C.c_void_p, C.c_void_p, _i32p, _f32p, C.c_int64, C.c_int64, C.c_int32]),
    "el_cml_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    "el_cml_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), _i32p, _i32p, _i32p, C.c_int64, C.c_float, C.c_float,
                                 _f32p, _f32p, _f64p]),
    "el_cml_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), _i32p, _i32p, _i32p, C.c_int64, C.c_float, C.c_float,
                               C.c_float, _f32p, _f32p, _f32p, _f32p, C.c_int64, _f64p, C.c_void_p, C.c_size_t]),
    "el_cml_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BprmfState), _i32p, _i32p, _i32p, C.c_int64, C.c_float,
                                    C.c_float, C.c_float, C.c_int32, C.c_float, _f64p, C.c_void_p, C.c_size_t]),
    "el_cml_prepare_items": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_int64, C.c_int32, _f32p, _f32p]),
    "el_cml_rescore": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, _f32p, _f32p, C.c_int32, _i32p, C.c_int64, C.c_int64, C.c_int32,
                                 C.c_int64, _f32p]),
    "el_pwmf_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    "el_pwmf_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(PwmfState), _i32p, _i32p, C.c_int64, _f32p]),
    "el_pwmf_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(PwmfState), _i32p, _i32p, _f32p, C.c_int64, C.c_int,
                                     C.c_int, C.c_int32, C.c_float, _f64p, C.c_void_p, C.c_size_t]),
    "el_pwmf_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(PwmfState), _i32p, _i32p, _f32p, C.c_int64, C.c_int64, C.c_int,
                                _f64p, C.c_void_p, C.c_size_t]),
    "el_pwmf_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(PwmfState), C.c_int, C.c_int, C.c_int32, C.c_float]),
    "el_pwmf_train_loop_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "el_pwmf_train_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(PwmfState), _i64p, _i32p, C.c_void_p, C.c_uint64, C.c_uint64,
                                     C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int32, C.c_void_p, _f64p, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_size_t]),
    "el_pwmf_link_values": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, C.c_int64, C.c_int64, C.c_int32, C.c_int, _f32p,
                                      C.c_int64]),
    "el_dense_topk": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                _i64p, _i32p, _i64p, _i32p, C.c_int32, _i32p, _f32p]),
}

_lib = None


class ElliotHipError(RuntimeError):
    pass


def load():
    """Load libelliot_hip.so (once).  Raises ElliotHipError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ElliotHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  elliot_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().el_last_error()
        raise ElliotHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
