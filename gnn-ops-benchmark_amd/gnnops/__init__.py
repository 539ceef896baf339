"""gnnops — MI355X (gfx950) kernels behind the gnn-ops-benchmark op API.

``import gnnops`` exposes the reference's op signatures on ROCm device tensors (see ops.py).
``gnnops.install()`` additionally routes the ATen ops the reference scripts call by name
(torch.index_select, Tensor.index_add_, torch.gather, Tensor.scatter_add_) to these kernels, and the
sibling packages ``torch_scatter`` / ``torch_sparse`` in this directory provide the import seam
(``from torch_scatter import scatter_add``) — so an unchanged ``op_bm_scripts/benchmark_*.py`` runs on them.
"""
from ._lib import GnnopsError, LIB_PATH, load as load_library
from .ops import (
    Plan,
    index_add_select_sum,
    clear_plan_cache,
    get_plan,
    index_add,
    index_add_,
    index_max,
    index_select_sum,
    scatter_add_,
    scatter_reduce_mul_,
    set_plan_cache,
)
# the differentiable front ends (they ARE the raw ops when nothing requires grad; gnnops.ops.* are the raw forms)
from .autograd import (addmm, gather, index_select, matmul, scatter, scatter_add, scatter_max, scatter_mean, scatter_min,
                       scatter_mul, scatter_sum)
from .sparse import (coalesce, coalesce_sparse_tensor, sddmm, sort, sparse_mm, spmm, spmm_csr, spmm_t, spspmm, transpose,
                     transpose_contiguous)
from .segment import (expand_rowptr, gather_coo, gather_csr, rowptr_from_sorted, scatter_log_softmax, scatter_logsumexp, scatter_softmax,
                      scatter_std, segment_coo, segment_csr)
from . import autograd, layers
from .aten import install, uninstall, installed

__all__ = [
    "GnnopsError", "LIB_PATH", "load_library", "Plan", "clear_plan_cache", "gather", "get_plan", "index_add",
    "index_add_", "index_max", "index_select", "index_select_sum", "scatter", "scatter_add", "scatter_add_",
    "scatter_max", "scatter_mean", "scatter_min", "scatter_mul", "scatter_reduce_mul_", "scatter_sum",
    "set_plan_cache", "install", "uninstall", "installed", "coalesce", "coalesce_sparse_tensor", "sort", "sparse_mm",
    "spmm", "spmm_csr", "spmm_t", "sddmm", "expand_rowptr", "spspmm", "transpose", "transpose_contiguous", "addmm", "matmul", "index_add_select_sum", "segment_csr", "segment_coo", "gather_csr", "gather_coo",
    "rowptr_from_sorted", "scatter_softmax", "scatter_log_softmax", "scatter_logsumexp", "scatter_std", "autograd", "layers",
]
