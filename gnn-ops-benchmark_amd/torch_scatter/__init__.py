"""Import seam for the reference scripts: ``from torch_scatter import scatter_add`` etc.
(op_bm_scripts/benchmark_scatter_add.py:5-7, benchmark_scatter_min.py:5-7, ...).

The real torch_scatter (pinned 2.0.9, requirements.txt:212) is a CUDA extension that is not
installed on the MI355X boxes; this package exports the same names with the same signatures, backed
by the gfx950 kernels in gnnops. No CPU path: CPU tensors raise.
"""
from gnnops.autograd import scatter
from gnnops.segment import (
    gather_coo,
    gather_csr,
    scatter_log_softmax,
    scatter_logsumexp,
    scatter_softmax,
    scatter_std,
    segment_coo,
    segment_csr,
)


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "mean")


def scatter_mul(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "mul")


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "min")


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "max")

__version__ = "2.0.9+gnnops.gfx950"
__all__ = ["scatter", "scatter_add", "scatter_sum", "scatter_mean", "scatter_min", "scatter_max", "scatter_mul",
           "segment_csr", "segment_coo", "gather_csr", "gather_coo", "scatter_softmax", "scatter_log_softmax",
           "scatter_logsumexp", "scatter_std"]
