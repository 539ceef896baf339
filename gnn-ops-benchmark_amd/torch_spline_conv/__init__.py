"""Import seam: ``from torch_spline_conv import spline_conv`` (ops.txt:17-19, 29-31) resolves to the MI355X kernels
(gnnops/spatial.py, csrc/spline.hip). The real package is absent on both boxes, so the name does not collide."""
from gnnops.spatial import spline_basis, spline_conv, spline_weighting

__version__ = "1.2.1+gnnops"
__all__ = ["spline_basis", "spline_weighting", "spline_conv"]
