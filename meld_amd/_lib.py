"""ctypes binding of libmeld_hip.so (the C-ABI declared in include/meld_hip.h).

The product path has no CPU fallback: if the shared library is missing or fails to load,
``get_lib()`` raises and every graph / filter operation fails loudly.
"""
from __future__ import annotations

from ._options import is_set, opt

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libmeld_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

_i32, _i64, _f64, _sz, _ptr = C.c_int, C.c_int64, C.c_double, C.c_size_t, C.c_void_p

# name -> (restype, argtypes); mirrors include/meld_hip.h one to one
SIGNATURES = {
    "meld_abi_version": (_i32, []),
    "meld_last_error": (C.c_char_p, []),
    "meld_device_count": (_i32, []),
    "meld_knn_padded_dim": (_i32, [_i32]),
    "meld_knn_tile_refs": (_i32, []),
    "meld_knn_block_queries": (_i32, []),
    "meld_knn_row_capacity": (_i32, [_i32]),
    "meld_col_sums_f64": (_i32, [_ptr, _i64, _i32, _ptr, _ptr]),
    "meld_col_stats_temp_bytes": (_sz, [_i32]),
    "meld_col_stats_f64": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    "meld_knn_prepare_refs": (_i32, [_ptr, _i64, _i32, _ptr, _i32, _ptr, _ptr, _ptr, _ptr]),
    "meld_knn_prepare_queries": (_i32, [_ptr, _i64, _i32, _ptr, _i32, _i64, _i64, _ptr, _ptr]),
    "meld_knn_topk": (_i32, [_ptr, _ptr, _i64, _i32, _i64, _i32, _ptr, _ptr, _ptr, _ptr]),
    "meld_knn_refine": (
        _i32,
        [_ptr, _i64, _i32, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _f64, _f64, _ptr, _f64, _ptr, _f64, _ptr, _ptr, _ptr,
         _ptr, _ptr, _ptr, _i32, _ptr, _f64, _ptr, _i32, _ptr],
    ),
    "meld_knn_error_coef": (_f64, [_i32]),
    "meld_knn16_kblocks": (_i32, [_i32]),
    "meld_knn16_tile_refs": (_i32, []),
    "meld_knn16_split_dims": (_i32, [_i32]),
    "meld_knn16_block_queries": (_i32, []),
    "meld_knn16_row_capacity": (_i32, [_i32]),
    "meld_knn16_error_coef": (_f64, [_i32, _i32]),
    "meld_knn16_error_coef_const": (_f64, [_i32, _i32]),
    "meld_knn16_error_coef_lin": (_f64, [_i32]),
    "meld_knn16_resident_blocks": (_i32, [_i32, _i32]),
    "meld_knn16_tile_bytes": (_sz, [_i32]),
    "meld_knn16_query_bytes": (_sz, [_i32]),
    "meld_knn16_prepare": (_i32, [_ptr, _i64, _i32, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "meld_knn16_prepare_scaled": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "meld_knn16_prepare_cross": (_i32, [_ptr, _i64, _i64, _i32, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "meld_knn16_prepare_rows": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr]),
    "meld_knn16_research_thresholds": (_i32, [_ptr, _i64, _i64, _ptr, _ptr, _i32, _i32, _ptr, _ptr, _f64, _f64, _f64, _ptr, _ptr, _ptr]),
    "meld_knn16_bounds_bytes": (_sz, [_i64, _i64]),
    "meld_knn16_bounds_temp_bytes": (_sz, [_i64, _i32, _i64]),
    "meld_knn16_bounds": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _i32, _ptr, _ptr, _ptr]),
    "meld_knn16_bounds_from_spheres": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _i32, _ptr, _ptr, _ptr]),
    "meld_knn16_tile_spheres": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _i64, _i64, _ptr]),
    "meld_knn16_sphere_layout": (_i32, [_i64, _i32, _ptr, _ptr]),
    "meld_knn16_topk": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i32, _i64, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _i32, _f64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "meld_knn16_block_work": (_i32, [_ptr, _ptr, _i64, _i32, _i64, _i32, _ptr, _ptr, _ptr, _ptr]),
    "meld_knn16_step_lists": (_i32, [_ptr, _ptr, _i64, _i32, _i64, _i32, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr]),
    "meld_knn16_list_scratch_bytes": (_sz, [_i64]),
    "meld_knn16_step_lists_direct": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _i64, _ptr, _ptr]),
    "meld_knn16_step_lists_direct_lead": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _i64, _ptr, _i32, _ptr]),
    "meld_knn16_topk_listed": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i32, _i64, _i32, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i32, _f64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr]),
    "meld_knn16_topk_listed_partial": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i32, _i64, _i32, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i32, _f64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "meld_knn16_partial_filter": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr]),
    "meld_knn16_seed_thresholds": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _i64, _i64, _i32, _f64, _i32, _ptr, _ptr]),
    "meld_knn16_seed_thresholds_mfma": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i32, _i64, _i64, _i32, _f64, _i32, _i32, _ptr, _ptr]),
    "meld_knn16_max_slices": (_i32, [_i32]),
    "meld_knn16_merge_slices": (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _i32, _ptr, _ptr, _ptr, _ptr]),
    "meld_knn_radius_exact": (
        _i32,
        [_ptr, _i64, _i32, _i64, _ptr, _i32, _ptr, _i32, _f64, _f64, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f64, _ptr],
    ),
    "meld_knn_pair_distances": (_i32, [_ptr, _i32, _ptr, _ptr, _i64, _i32, _ptr, _ptr]),
    "meld_scan_temp_bytes": (_sz, [_i64]),
    "meld_exclusive_scan_i32_i64": (_i32, [_ptr, _ptr, _i64, _ptr, _sz, _ptr]),
    "meld_coo_emit": (
        _i32,
        [_i64, _i64, _ptr, _ptr, _ptr, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr],
    ),
    "meld_sort_temp_bytes": (_sz, [_i64]),
    "meld_sort_pairs_u64_f64": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i32, _ptr, _sz, _ptr]),
    "meld_merge_temp_bytes": (_sz, [_i64]),
    "meld_coo_merge": (_i32, [_ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    "meld_csr_from_keys": (_i32, [_ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr]),
    "meld_csr_bucket_slots": (_i32, []),
    "meld_coo_scatter_rows": (_i32, [_ptr, _ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr]),
    "meld_coo_emit_scatter": (_i32, [_i64, _ptr, _ptr, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr]),
    "meld_coo_partition_remote": (_i32, [_ptr, _ptr, _i64, _i64, _i32, _i32, _i64, _ptr, _ptr, _ptr]),
    "meld_csr_rows_sort_merge": (_i32, [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _f64, _ptr]),
    "meld_csr_compact_rows": (_i32, [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "meld_csr_compact_rows_sums": (_i32, [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _f64, _ptr, _ptr]),
    "meld_csr_row_sums": (_i32, [_ptr, _ptr, _i64, _f64, _ptr, _ptr]),
    "meld_csr_anisotropy": (_i32, [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _f64, _ptr]),
    "meld_csr_anisotropy_degrees": (_i32, [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _f64, _ptr, _ptr]),
    "meld_spmm_dot_slots": (_i32, []),
    "meld_cheby_step": (
        _i32,
        [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i32, _ptr, _i64, _ptr, _ptr, _ptr, _f64, _f64, _f64, _f64, _ptr, _ptr],
    ),
    "meld_lanczos_steps": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr, _ptr]),
    "meld_lanczos_spmv": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "meld_lanczos_alpha": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _ptr]),
    "meld_lanczos_axpy": (_i32, [_ptr, _ptr, _i64, _ptr, _ptr, _ptr]),
    "meld_lanczos_beta": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _ptr]),
    "meld_lanczos_fold": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _ptr]),
    "meld_lanczos_axpy3": (_i32, [_ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr]),
    "meld_pt_geometry": (_i32, [_ptr, _ptr, _ptr, _ptr]),
    "meld_pt_num_blocks": (_i32, [_i64]),
    "meld_pt_seg_len": (_i64, [_i32]),
    "meld_pt_debug_ablate": (_i32, [_i32]),
    "meld_pt_debug_stamps": (_i32, [_ptr]),
    "meld_pt_stream_len": (_i64, [_i64, _i32]),
    "meld_pt_desc_len": (_i64, [_i32]),
    "meld_pt_build": (_i32, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i32, _ptr, _ptr, _ptr, _ptr]),
    "meld_cheby_step_wide": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i32, _ptr, _i64, _ptr, _ptr, _f64, _f64, _f64, _ptr]),
    "meld_pt_cheby_step": (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _ptr, _i64, _ptr, _ptr, _ptr, _f64, _f64, _f64, _f64, _ptr, _ptr]),
    "meld_pt_cheby_run": (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _ptr, _ptr, _ptr, _ptr, _i32, _f64, _f64, _ptr, _ptr]),
    "meld_pt_lanczos_steps": (_i32, [_ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr, _ptr, _ptr]),
    "meld_pt_lanczos_spmv": (_i32, [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "meld_kmeans_max_blocks": (_i32, []),
    "meld_kmeans_assign": (_i32, [_ptr, _i64, _i32, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _i32, _ptr]),
    "meld_scale_f64": (_i32, [_ptr, _f64, _ptr, _i64, _ptr]),
    "meld_axpby_f64": (_i32, [_f64, _ptr, _f64, _ptr, _i64, _ptr, _ptr]),
    "meld_normalize_rows_l1": (_i32, [_ptr, _ptr, _i64, _i32, _ptr]),
    "meld_assign_nearest": (_i32, [_ptr, _i64, _i32, _ptr, _i32, _ptr, _ptr, _ptr, _ptr]),
    "meld_chain_order": (_i32, [_ptr, _i64, _i32, _i32, _ptr, _ptr]),
    "meld_gather_rows_f64": (_i32, [_ptr, _ptr, _i64, _i32, _ptr, _ptr]),
    "meld_frame_max_dims": (_i32, []),
    "meld_cov_sample_f64": (_i32, [_ptr, _i64, _i32, _ptr, _i64, _ptr, _ptr]),
    "meld_rotate_rows_f64": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _ptr]),
    "meld_argsort_u32_temp_bytes": (_sz, [_i64]),
    "meld_argsort_u32": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _ptr, _sz, _ptr]),
    "meld_order_starts": (_i32, [_ptr, _i64, _i32, _ptr, _ptr]),
    "meld_order_pick_centroids": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _i32, _i32, _ptr, _ptr]),
    "meld_order_update_keys": (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _ptr]),
    "meld_rccl_available": (_i32, []),
    "meld_rccl_unique_id": (_i32, [_ptr]),
    "meld_rccl_comm_create": (_i32, [_ptr, _i32, _i32, _ptr]),
    "meld_rccl_comm_destroy": (_i32, [_ptr]),
    "meld_rccl_all_gather": (_i32, [_ptr, _ptr, _ptr, _sz, _ptr]),
    "meld_rccl_all_reduce_sum_f64": (_i32, [_ptr, _ptr, _sz, _ptr]),
    "meld_cheby_run_sharded": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i32, _ptr, _ptr, _ptr, _ptr, _i32, _f64, _f64, _ptr, _ptr]),
    "meld_lanczos_steps_sharded": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "meld_factorize_max_groups": (_i32, []),
    "meld_factorize_max_words": (_i32, []),
    "meld_factorize_temp_bytes": (_sz, [_i64]),
    "meld_factorize_labels": (_i32, [_ptr, _i64, _i32, _ptr, _sz, _ptr, _ptr]),
    "meld_factorize_codes": (_i32, [_ptr, _i64, _ptr, _ptr, _ptr]),
    "meld_indicator_signal": (_i32, [_ptr, _ptr, _ptr, _i64, _i64, _i32, _ptr, _ptr]),
    "meld_scatter_rows_f64": (_i32, [_ptr, _ptr, _i64, _i32, _ptr, _ptr]),
}



class PtLayout(C.Structure):
    """``meld_pt_layout_t`` of include/meld_hip.h (device pointers of the panel-tiled copy of W)."""

    _fields_ = [("blk_row", _ptr), ("blk_ntile", _ptr), ("blk_ndist", _ptr), ("seg", _ptr), ("list_cols", _ptr),
                ("pval", _ptr), ("pidx", _ptr), ("nb", C.c_int32), ("pval32", _ptr), ("stream_len", C.c_int64), ("cdesc", _ptr)]


_lib = None


class MeldHipError(RuntimeError):
    """A libmeld_hip.so entry point returned a non-zero status."""


def get_lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = opt("MELD_HIP_LIB") or LIB_PATH  # (development: an alternative build of the same library)
    if not os.path.exists(path):
        raise ImportError(
            "meld_amd: {} not found. The MI355X HIP extension is mandatory (there is no CPU "
            "fallback); build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python -m meld_amd.build`.".format(path)
        )
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud failure
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = get_lib().meld_last_error()
        raise MeldHipError("{} failed with status {}: {}".format(what or "libmeld_hip call", status, (msg or b"").decode()))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
