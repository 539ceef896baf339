"""ctypes binding of the C ABI in include/gnnops.h (csrc/libgnnops.so, built for gfx950).

There is no CPU or eager fallback: if the HIP library is missing this module raises at import of the
first op, and every op raises on non-GPU tensors.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GNNOPS_LIB_PATH: another build of the same ABI (A/B timing of two builds on one box: tools/time_partition.py)
LIB_PATH = os.environ.get("GNNOPS_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libgnnops.so")

ABI_VERSION = 2   # GNNOPS_ABI_VERSION of include/gnnops.h
F32, F16, BF16 = 0, 1, 2
SUM, MEAN, MIN, MAX, MUL = 0, 1, 2, 3, 4
OK, EINVAL, EWORKSPACE, ELAUNCH, EUNSUPPORTED = 0, 1, 2, 3, 4  # status codes of include/gnnops.h
REDUCE_CODE = {"sum": SUM, "add": SUM, "mean": MEAN, "min": MIN, "max": MAX, "mul": MUL}

# name -> (restype, argtypes); must list every symbol include/gnnops.h declares (tests check this).
_i64, _vp, _ci, _sz = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
SIGNATURES = {
    "gnnops_version": (_ci, []),
    "gnnops_last_error": (ctypes.c_char_p, []),
    "gnnops_diag_stream_mix": (_ci, [_vp, _vp, _i64, _ci, _vp]),
    "gnnops_index_max": (_ci, [_vp, _i64, _vp, _vp]),
    "gnnops_plan_workspace_bytes": (_sz, [_i64, _i64]),
    "gnnops_plan_build": (_ci, [_vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "gnnops_plan_small_fits": (_ci, [_i64, _i64]),
    "gnnops_plan_build_small": (_ci, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gnnops_segment_reduce": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _ci, _ci, _vp]),
    "gnnops_bucket_workspace_bytes": (_sz, [_i64, _i64]),
    "gnnops_bucket_partition": (_ci, [_vp, _i64, _i64, _vp, _sz, _vp]),
    "gnnops_bucket_partition_window": (_ci, [_vp, _i64, _i64, _i64, _vp, _sz, _vp]),
    "gnnops_bucket_layout": (_ci, [_i64, _i64, _vp, _vp, _vp]),
    "gnnops_bucket_reduce": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _ci, _ci, _ci, _vp]),
    "gnnops_bucket_reduce_hubs": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _ci, _ci, _ci, _vp, _sz, _vp]),
    "gnnops_hub_workspace_bytes": (_sz, [_i64, _i64, _ci]),
    "gnnops_segment_reduce_hubs": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _ci, _ci, _vp, _sz, _vp]),
    "gnnops_bucket_select_hubs": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _ci, _vp, _sz, _vp]),
    "gnnops_index_select_planned_hubs": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp, _sz, _vp]),
    "gnnops_bucket_select": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _ci, _vp]),
    "gnnops_scatter_rows_oneshot": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _ci, _ci, _ci, _vp, _sz, _vp]),
    "gnnops_scatter_elementwise_workspace_bytes": (_sz, [_i64, _i64, _i64, _ci, _ci]),
    "gnnops_scatter_elementwise": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _ci, _ci, _vp, _sz, _vp]),
    "gnnops_scatter_elementwise_ix": (_ci, [_vp, _vp, _ci, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _ci, _ci, _vp, _sz, _vp]),
    "gnnops_scatter1d_workspace_bytes": (_sz, [_i64, _i64]),
    "gnnops_scatter1d_minmax": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _ci, _ci, _vp, _sz, _vp]),
    "gnnops_scatter1d_sum": (_ci, [_vp, _vp, _vp, _i64, _i64, _ci, _ci, _vp, _sz, _vp]),
    "gnnops_scatter_elementwise_ixa": (_ci, [_vp, _vp, _ci, _vp, _vp, _ci, _i64, _i64, _i64, _i64, _ci, _ci, _ci, _vp, _sz, _vp]),
    "gnnops_transpose2d_cvt": (_ci, [_vp, _vp, _i64, _i64, _ci, _vp]),
    "gnnops_transpose2d_cvt_max": (_ci, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "gnnops_narrow_index": (_ci, [_vp, _vp, _i64, _ci, _i64, _vp]),
    "gnnops_index_select": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp]),
    "gnnops_index_select_planned": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp]),
    "gnnops_gather": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp]),
    "gnnops_fused_select_sum_workspace_bytes": (_sz, []),
    "gnnops_fused_index_select_sum": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp, _sz, _vp]),
    "gnnops_spmm": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp]),
    "gnnops_spmm_hubs": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp, _sz, _vp]),
    "gnnops_permute": (_ci, [_vp, _vp, _vp, _i64, _ci, _vp]),
    "gnnops_sort_workspace_bytes": (_sz, [_i64, _i64, _i64, _ci]),
    "gnnops_sort": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _ci, _ci, _vp, _sz, _vp]),
    "gnnops_sort_rows_max_len": (_i64, []),
    "gnnops_sort_rows2_max_len": (_i64, []),
    "gnnops_sort_rows_f32": (_ci, [_vp, _vp, _vp, _i64, _i64, _ci, _vp]),
    "gnnops_sort_rows_f32_i32": (_ci, [_vp, _vp, _vp, _i64, _i64, _ci, _vp]),
    "gnnops_sort_rows2_f32": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _ci, _vp]),
    "gnnops_sort_rows2_f32_i32": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _ci, _vp]),
    "gnnops_coalesce_workspace_bytes": (_sz, [_i64]),
    "gnnops_coalesce": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gnnops_transpose2d": (_ci, [_vp, _vp, _i64, _i64, _ci, _vp]),
    "gnnops_transpose_batched": (_ci, [_vp, _vp, _i64, _i64, _i64, _ci, _vp]),
    "gnnops_spspmm_workspace_bytes": (_sz, [_i64]),
    "gnnops_spspmm_count": (_ci, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "gnnops_spspmm_expand": (_ci, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp]),
    "gnnops_rowptr_workspace_bytes": (_sz, [_i64]),
    "gnnops_rowptr_from_sorted": (_ci, [_vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "gnnops_owner_counts": (_ci, [_vp, _i64, _i64, _ci, _vp, _vp]),
    "gnnops_rowptr_expand": (_ci, [_vp, _i64, _i64, _vp, _vp]),
    "gnnops_sddmm": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _ci, _vp]),
    "gnnops_edge_reduce": (_ci, [_ci, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _ci, _vp, _ci,
                                 ctypes.c_float, ctypes.c_float, _ci, _vp]),
    "gnnops_edge_reduce_hub_workspace_bytes": (_sz, [_i64, _i64]),
    "gnnops_edge_reduce_hubs": (_ci, [_ci, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _ci, _vp,
                                      _ci, ctypes.c_float, ctypes.c_float, _ci, _vp, _sz, _vp]),
    "gnnops_edge_grad": (_ci, [_ci, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _ci, _vp]),
    "gnnops_spline_basis": (_ci, [_vp, _vp, _vp, _i64, _ci, _ci, _vp, _vp, _ci, _vp]),
    "gnnops_spline_weighting": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp]),
    "gnnops_spline_conv": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _ci, _vp]),
    "gnnops_grid_cluster": (_ci, [_vp, _i64, _ci, _vp, _vp, _vp, _vp, _ci, _vp]),
    "gnnops_knn": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _ci, _ci, _ci, _vp, _ci, _vp]),
    "gnnops_knn_grid_cells": (_ci, [_vp, _i64, _ci, _ci, _vp, _vp, _vp]),
    "gnnops_knn_grid_query": (_ci, [_vp, _vp, _i64, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]),
    "gnnops_radius_grid_query": (_ci, [_vp, _vp, _i64, _ci, ctypes.c_double, _ci, _ci, _vp, _vp, _vp, _vp, _vp]),
    "gnnops_radius": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _ci, ctypes.c_double, _ci, _vp, _ci, _vp]),
    "gnnops_fps": (_ci, [_vp, _vp, _vp, _vp, _i64, _ci, _vp, _vp, _ci, _vp]),
    "gnnops_random_walk": (_ci, [_vp, _vp, _vp, _i64, _ci, ctypes.c_uint64, _vp, _vp]),
    "gnnops_random_walk_node2vec": (_ci, [_vp, _vp, _vp, _i64, _ci, ctypes.c_double, ctypes.c_double, ctypes.c_uint64, _vp, _vp]),
    "gnnops_graclus_rounds": (_ci, [_vp, _vp, _vp, _i64, ctypes.c_uint64, _ci, _vp, _vp, _vp, _ci, _ci, _vp]),
    "gnnops_segment_composite": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _ci, ctypes.c_double, _vp]),
    "gnnops_segment_composite_hubs": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _ci, ctypes.c_double, _vp, _sz, _vp]),
    "gnnops_addmm_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gnnops_addmm": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _ci, _vp, _sz, _vp]),
    "gnnops_addmm_ld": (_ci, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _ci, _vp, _sz, _vp]),
    "gnnops_fused_index_add_select_sum_workspace_bytes": (_sz, [_i64, _i64]),
    "gnnops_fused_index_add_select_sum": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _vp, _sz, _vp]),
}

_lib = None


class GnnopsError(RuntimeError):
    pass


def load():
    """Load libgnnops.so; raises ImportError (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"gnnops: HIP library not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C {os.path.dirname(LIB_PATH)}`. There is no CPU fallback."
        )
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if L.gnnops_version() != ABI_VERSION:     # GNNOPS_LIB_PATH may point at another build: it must speak this binding's ABI
        raise ImportError(f"gnnops: {LIB_PATH} reports ABI version {L.gnnops_version()}, this binding needs {ABI_VERSION}; rebuild it")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = load().gnnops_last_error().decode("utf-8", "replace")
        if rc == EUNSUPPORTED:
            raise NotImplementedError(f"gnnops.{what}: {msg}")
        raise GnnopsError(f"gnnops.{what} failed (code {rc}): {msg}")
