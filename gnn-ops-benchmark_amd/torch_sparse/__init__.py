"""Import seam for `from torch_sparse import coalesce` (op_bm_scripts/benchmark_sparse_coalesce.py:7) and the
torch_sparse ops BASELINE.json's north_star names: spmm, spspmm, transpose.

The real torch_sparse (pinned 0.6.12, requirements.txt:213) is not installed on the MI355X boxes; this
package exports the same names and signatures on top of the gfx950 kernels. No CPU path."""
from gnnops.sparse import coalesce, spmm, spspmm, transpose

__version__ = "0.6.12+gnnops.gfx950"
__all__ = ["coalesce", "spmm", "spspmm", "transpose"]
