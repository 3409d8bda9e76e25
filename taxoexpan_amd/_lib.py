"""ctypes binding of libtxe.so (include/txe.h).  There is NO fallback: if the HIP library cannot be loaded the
first compute call raises -- a GPU box must never silently run something else."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtxe.so")
TAIL_CHAIN_BYTES = 1024        # TXE_TAIL_CHAIN_BYTES of include/txe.h

P, I, L, F, D, U64, SZ = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double, C.c_ulonglong, C.c_size_t

# name -> (restype, argtypes); mirrors include/txe.h one to one (tests/test_cabi.py checks the header against this)
SIGNATURES = {
    "txe_dropout_mask_bytes": (SZ, [L, I]),
    "txe_dropout_mask": (I, [L, I, F, U64, P, P]),
    "txe_gat_padded_k": (I, [I, I]),
    "txe_gat_padded_f": (I, [I, I]),
    "txe_gat_pack_weights": (I, [P, P, P, I, I, I, P, P]),
    "txe_gat_build_x": (I, [P, L, I, I, P, P, I, P, P]),
    "txe_gat_layer_prepare": (I, [P, L, I, I, P, P, I, P, P, P, P, I, I, P, F, U64, P, P]),
    "txe_gcn_layer_prepare": (I, [P, L, I, I, P, P, I, P, P, I, P, F, U64, P, I, P, P]),
    "txe_gcn_layers_prepare": (I, [P, I, P, I, P, P]),
    "txe_gather_add_rows": (I, [P, L, P, P, L, P, L, I, P, L, P]),
    "txe_gat_dense_ws_bytes": (SZ, [I, I, I, I, I, I]),
    "txe_gat_dense_fwd": (I, [P, I, I, I, P, I, I, F, P, P, P, SZ, P]),
    "txe_gat_dense_split_ws_bytes": (SZ, [I, I, I, I, I]),
    "txe_gat_dense_split_xt_bytes": (SZ, [I, I, I, I, I]),
    "txe_gat_dense_bwd_split_ws_bytes": (SZ, [I, I, I, I, I]),
    "txe_gat_dense_fwd_split": (I, [P, I, I, I, P, I, I, P, P, P, P, P, SZ, P]),
    "txe_gat_dense_bwd": (I, [P, I, I, I, P, I, P, P, P, P, I, I, F, P, P, I, I, F, P, P, P, P, P, I, P, I, P, P, SZ, P]),
    "txe_gat_tail_flush": (I, [P, P]),
    "txe_zero_cols": (I, [P, L, I, I, I, P]),
    "txe_gat_dx_streams": (I, [I, I, I]),
    "txe_gat_aggregate_fwd": (I, [P, P, I, P, L, P, P, I, I, I, F, F, U64, I, F, P, L, P, P, I, P, F, P, I, P]),
    "txe_gat_aggregate_table_supported": (I, [I, I, L, I, I]),
    "txe_gat_aggregate_table_fwd": (I, [P, P, I, P, L, P, P, P, I, I, I, F, I, F, P, L, P, I, P, I, P]),
    "txe_gat_aggregate_bwd": (I, [P, P, P, P, P, I, P, L, P, P, I, I, I, F, F, U64, P, P, L, P, L, P, P, I, P, I, P]),
    "txe_leaky_relu_bwd": (I, [P, P, F, L, P, P]),
    "txe_head_mean_fwd": (I, [P, I, I, L, P, P]),
    "txe_head_mean_bwd": (I, [P, I, I, L, P, P]),
    "txe_gcn_padded_f": (I, [I]),
    "txe_gcn_pack_weights": (I, [P, I, I, P, P]),
    "txe_gcn_dense_ws_bytes": (SZ, [I, I, I, I, I]),
    "txe_gcn_dense_fwd": (I, [P, I, I, I, P, I, F, P, P, P, SZ, P]),
    "txe_gcn_dense_bwd": (I, [P, I, I, I, P, I, P, I, F, P, P, I, I, F, P, P, P, I, P, SZ, P]),
    "txe_gcn_norm": (I, [P, I, P, P]),
    "txe_gcn_aggregate_fwd": (I, [P, P, I, P, L, P, P, I, F, I, P, L, P]),
    "txe_gcn_aggregate_bwd_ws_bytes": (SZ, [I, I]),
    "txe_gcn_aggregate_bwd": (I, [P, P, I, P, L, P, I, P, L, P, P, SZ, P]),
    "txe_readout_fwd": (I, [P, I, P, L, P, P, I, P, P, P]),
    "txe_readout_bwd": (I, [P, I, P, L, P, P, I, I, P, P, P, P, L, P, P, P]),
    "txe_readout_multi_fwd": (I, [P, I, P, L, P, I, I, P, P, P]),
    "txe_readout_multi_bwd": (I, [P, I, P, I, I, P, P, P, L, P]),
    "txe_linear_fwd": (I, [P, L, I, P, L, I, I, P, P, I, I, P, P]),
    "txe_linear_bwd_ws_bytes": (SZ, [I, I, I, I]),
    "txe_linear_bwd": (I, [P, L, I, P, L, I, I, P, I, I, P, P, P, L, P, L, P, P, P, SZ, P]),
    "txe_bilinear_project": (I, [P, L, I, I, P, I, P, L, P, SZ, P]),
    "txe_bilinear_pair_fwd": (I, [P, L, P, L, I, I, I, P, I, P, P, P]),
    "txe_bilinear_query_fwd": (I, [P, L, P, L, I, I, I, P, I, P, P, P]),
    "txe_bilinear_query_project": (I, [P, L, I, I, I, P, P, P]),
    "txe_bilinear_query_dot": (I, [P, L, P, I, I, I, P, P]),
    "txe_bilinear_query_bwd_ws_bytes": (SZ, [I, I, I]),
    "txe_bilinear_query_bwd": (I, [P, L, P, L, I, I, I, I, P, P, P, P, L, P, P, SZ, P]),
    "txe_bilinear_runs_fwd": (I, [P, L, P, L, P, I, I, I, I, P, I, P, P, P]),
    "txe_bilinear_runs_bwd_ws_bytes": (SZ, [I, I, I]),
    "txe_bilinear_runs_bwd": (I, [P, L, P, L, P, I, I, I, I, I, P, P, P, P, L, P, P, SZ, P]),
    "txe_rows_find_runs": (I, [P, L, I, I, P, P, P, P]),
    "txe_score_topk_tiles": (I, [I]),
    "txe_score_topk_block": (I, [P, L, I, P, L, I, I, I, I, I, I, P, P, P, P, P, P, SZ, P, P]),
    "txe_topk_merge": (I, [P, P, I, L, I, I, P, P, P]),
    "txe_bilinear_stacked_fwd": (I, [P, L, P, L, P, P, I, I, I, P, I, P, P, P]),
    "txe_bilinear_stacked_bwd_ws_bytes": (SZ, [I, I, I]),
    "txe_bilinear_stacked_bwd": (I, [P, L, P, L, P, P, I, I, I, I, P, P, P, P, L, P, P, SZ, P]),
    "txe_bilinear_folded_fwd": (I, [P, L, I, I, P, L, I, P, L, I, P, P, I, I, P, I, P, P, P, I, I, I, P]),
    "txe_runs_expand": (I, [P, I, I, P, P]),
    "txe_bilinear_folded_bwd": (I, [P, L, I, I, P, L, I, P, L, I, P, P, I, I, I, P, P, P, P, P, L, P, P, P, P, I, I, P]),
    "txe_bilinear_pair_bwd_ws_bytes": (SZ, [I, I, I]),
    "txe_bilinear_pair_bwd": (I, [P, L, P, L, I, I, I, P, I, P, P, P, P, L, P, L, P, P, SZ, P]),
    "txe_score_block": (I, [P, L, I, P, L, I, I, I, P, L, P, SZ, P, SZ, P, P]),
    "txe_score_split_ws_bytes": (SZ, [I, I, I]),
    "txe_score_count_block": (I, [P, L, I, P, L, I, I, I, P, P, I, P, P, SZ, P, P]),
    "txe_score_positives": (I, [P, L, I, P, L, I, I, I, P, P, P, SZ, P]),
    "txe_rank_finalize": (I, [P, I, P, P, I, P, P]),
    "txe_gemm_tail_ws_bytes": (SZ, []),
    "txe_gemm_plain_split_ws_bytes": (SZ, [I, I, I]),
    "txe_gat_collapse_split_ws_bytes": (SZ, [I, I, I, I]),
    "txe_gemm_plain": (I, [I, P, L, P, L, P, L, I, I, I, I, I, P, SZ, P]),
    "txe_build_csr_ws_bytes": (SZ, [I, I]),
    "txe_build_csr": (I, [P, P, I, I, P, P, P, P, P, P, P, SZ, P]),
    "txe_rank_block": (I, [P, L, I, I, P, P, P, I, P, P]),
    "txe_gat_collapse_ws_bytes": (SZ, [I, I, I, I, I, I, I]),
    "txe_gat_collapse_fwd": (I, [P, P, P, P, P, P, I, I, I, P, I, I, P, I, F, P, F, F, U64, P, P, P, I, P, P, P, P, P, P, L, P, P, P, P, SZ, P]),
    "txe_gat_collapse_e_tiles": (I, [I, I, I, I]),
    "txe_gat_collapse_fold_scores": (I, [P, I, I, I, I, P, P, P, F, I, I, P, P]),
    "txe_gat_collapse_bwd": (I, [P, P, P, P, P, P, I, I, I, P, I, I, P, I, P, P, P, P, I, F, P, F, F, U64, P, P, P, P, P, P, P, P, L, P, L, I, F,
                                 P, P, P, P, P, P, P, SZ, P]),
    "txe_gat_layers_prepare": (I, [P, I, P]),
    "txe_gat_fused_bwd_supported": (I, [I, I, I, I]),
    "txe_gat_collapse_bwd_fused_ws_bytes": (SZ, [I, I, I, I, I, I, I, I]),
    "txe_gat_collapse_bwd_fused": (I, [P, P, P, P, P, P, I, I, I, P, I, I, P, I, P, P, P, P, I, F, P, F, F, U64, P, P, P, P, P, P, P, P, L, P, L,
                                       F, P, L, I, I, F, F, U64, P, P, L, I, P, P, P, P, P, P, I, P, I, P, P, P, I, P, P, P, P, P, P, SZ, P]),
    "txe_gat_dense_fwd_split_src": (I, [P, L, P, P, P, F, I, I, I, P, I, I, P, P, P, SZ, P]),
    "txe_egonet_walk_plan_bytes": (SZ, [I]),
    "txe_egonet_walk_plan": (I, [P, P, P, P, P, P, I, I, P, P]),
    "txe_gcn_collapse_ws_bytes": (SZ, [I, I, I, I, I, I]),
    "txe_gcn_collapse_fwd": (I, [P, P, P, I, I, P, I, I, P, I, P, F, P, P, P, P, P, P, P, P, P, L, P, SZ, P]),
    "txe_gcn_collapse_bwd": (I, [P, P, P, I, I, P, I, I, P, I, P, I, F, P, P, P, P, P, P, P, P, L, I, F, P, P, P, P, P, I, P, SZ, P]),
    "txe_egonet_ws_bytes": (SZ, [I]),
    "txe_egonet_offsets": (I, [P, P, P, P, P, I, I, U64, I, P, P, SZ, P]),
    "txe_egonet_fill": (I, [P, P, P, P, P, P, I, I, U64, I, P, P, P, P, P, P, P, P, P, P]),
    "txe_info_nce": (I, [P, L, I, I, P, P, P, L, P]),
    "txe_adam_step": (I, [I, P, P, P, P, P, P, D, D, D, D, D, L, P]),
    "txe_dropout_uniform_host": (F, [U64, U64]),
    "txe_dropout_mask_word_host": (C.c_uint, [U64, U64, F]),
    "txe_profile_enable": (I, [I]),
    "txe_profile_reset": (I, []),
    "txe_profile_count": (I, []),
    "txe_profile_get": (I, [I, P, I, P, P, P]),
    "txe_profile_stream": (I, [I, P]),
    "txe_stream_order": (I, [P, P]),
    "txe_copy_stream": (I, [P, P, L, P]),
    "txe_split_packed_bytes": (SZ, [I, I]),
    "txe_split_pack": (I, [P, L, I, I, I, P, P]),
    "txe_split_packed_t_bytes": (SZ, [I, I]),
    "txe_split_pack_t": (I, [P, L, I, I, P, P]),
    "txe_gemm_tn_split": (I, [P, L, I, P, I, I, I, I, P, L, L, P]),
    "txe_gemm_nt_split": (I, [P, P, I, I, I, P, L, P]),
}

class GatPrepareDesc(C.Structure):
    """struct txe_gat_prepare_desc (include/txe.h)"""
    _fields_ = [("h", P), ("ld_h", L), ("n_nodes", I), ("Kh", I), ("pos", P), ("P", P), ("Pd", I), ("X", P), ("W", P), ("attn_l", P),
                ("attn_r", P), ("H", I), ("D", I), ("Wp", P), ("feat_drop_p", F), ("seed", U64), ("mask", P), ("x_dropped", I)]


class GcnPrepareDesc(C.Structure):
    """struct txe_gcn_prepare_desc (include/txe.h)"""
    _fields_ = [("h", P), ("ld_h", L), ("n_nodes", I), ("Kh", I), ("pos", P), ("P", P), ("Pd", I), ("X", P), ("W", P), ("Fo", I), ("Wp", P),
                ("drop_p", F), ("seed", U64), ("mask", P), ("x_dropped", I), ("bias_row", P)]


_ERR = {-1: "TXE_ERR_ARG", -2: "TXE_ERR_LAUNCH", -3: "TXE_ERR_WORKSPACE"}
VALUE_RETURNING = {"txe_gat_padded_k", "txe_gat_collapse_e_tiles", "txe_gat_padded_f", "txe_gcn_padded_f", "txe_profile_count", "txe_gat_fused_bwd_supported",
                   "txe_gat_aggregate_table_supported", "txe_gat_dx_streams", "txe_score_topk_tiles"}   # int results that are not status codes

_lib = None


class TxeError(RuntimeError):
    pass


def load():
    """dlopen libtxe.so and declare prototypes.  Raises if the library is absent (build it with
    `python taxoexpan_amd/csrc/build.py` or `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TxeError(f"HIP extension missing: {LIB_PATH} not built -- run taxoexpan_amd/csrc/build.py "
                       "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


_bound = {}


def call(name, *args):
    ent = _bound.get(name)
    if ent is None:                       # (bound function, "a non-zero int result is an error") looked up once per entry point
        ent = _bound[name] = (getattr(load(), name), SIGNATURES[name][0] is I and name not in VALUE_RETURNING)
    rc = ent[0](*args)
    if rc != 0 and ent[1]:
        raise TxeError(f"{name} failed: {_ERR.get(rc, rc)}")
    return rc


_pure = {}
_get_device = None


def _current_device():
    global _get_device
    if _get_device is None:
        import torch
        _get_device = torch._C._cuda_getDevice if torch.cuda.is_available() else (lambda: -1)
    return _get_device()


def pure(name, *args):
    """call() for the entry points that are pure functions of their integer arguments (and of the device's CU count): padded widths,
    workspace sizes, `*_supported` predicates -- the answer is cached (a ctypes call costs the host 2-4 us; a step asks ~10 of them)"""
    key = (name, _current_device()) + args              # (the plans depend on the device's CU count)
    v = _pure.get(key)
    if v is None:
        if len(_pure) > 4096:
            _pure.clear()
        v = _pure[key] = call(name, *args)
    return v


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """`with on_device(t.device):` = torch.cuda.device(dev) when `dev` is not the current device; nothing at all when it is (the usual
    case, one process per GPU: the guard object and two device switches cost ~8 us of host time per launch group)"""
    import torch
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def ptr(t):
    """device pointer of a (contiguous) tensor, or None"""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = None


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (the raw handle: no Stream object is built per launch)"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
