"""Import seam: ``from torch_cluster import knn_graph, fps, ...`` (ops.txt:33-41) resolves to the MI355X kernels
(gnnops/spatial.py, csrc/cluster.hip). The real package is absent on both boxes, so the name does not collide.
"""
from gnnops.spatial import fps, graclus_cluster, grid_cluster, knn, knn_graph, nearest, radius, radius_graph, random_walk

__version__ = "1.5.9+gnnops"
__all__ = ["graclus_cluster", "grid_cluster", "fps", "knn", "knn_graph", "radius", "radius_graph", "nearest", "random_walk"]
