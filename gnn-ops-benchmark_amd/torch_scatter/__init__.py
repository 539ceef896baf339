"""Import seam for the reference scripts: ``from torch_scatter import scatter_add`` etc.
(op_bm_scripts/benchmark_scatter_add.py:5-7, benchmark_scatter_min.py:5-7, ...).

The real torch_scatter (pinned 2.0.9, requirements.txt:212) is a CUDA extension that is not
installed on the MI355X boxes; this package exports the same names with the same signatures, backed
by the gfx950 kernels in gnnops. No CPU path: CPU tensors raise.
"""
from gnnops.ops import (
    scatter,
    scatter_add,
    scatter_max,
    scatter_mean,
    scatter_min,
    scatter_mul,
    scatter_sum,
)

__version__ = "2.0.9+gnnops.gfx950"
__all__ = ["scatter", "scatter_add", "scatter_sum", "scatter_mean", "scatter_min", "scatter_max", "scatter_mul"]
