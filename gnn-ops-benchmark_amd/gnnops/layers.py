"""Message-passing building blocks composed from the kernels (the single-layer forward passes
app_bm/groq_script.py:91-109 times — GIN / SAGE: gather neighbour rows, reduce per destination, one dense layer).

The unfused chain is index_select (writes [E, D]) followed by scatter_add (reads it back). `propagate_sum` is the fused
form: one spmm launch over the plan of the destination index reads each neighbour row once and writes each destination
row once — the [E, D] message tensor never exists.
"""
import torch

from . import autograd, sparse


def propagate_sum(x, edge_index, num_nodes=None):
    """out[i] = sum_{(j -> i) in edge_index} x[j]; edge_index int64 [2, E] = (source row j, destination row i)."""
    if num_nodes is None:
        num_nodes = x.size(0)
    return sparse.spmm_t(edge_index, None, x.size(0), num_nodes, x)  # plan cached under the edge_index tensor


def propagate_mean(x, edge_index, num_nodes=None):
    if num_nodes is None:
        num_nodes = x.size(0)
    s = propagate_sum(x, edge_index, num_nodes)
    deg = autograd.scatter(torch.ones(edge_index.size(1), 1, dtype=x.dtype, device=x.device), edge_index[1], 0, None, num_nodes, "sum")
    return s / deg.clamp_(min=1)


def gin_conv(x, edge_index, weight, bias=None, eps=0.0):
    """GIN layer with a single linear map: ((1 + eps) * x + sum_j x_j) @ weight + bias  (weight [D_in, D_out])."""
    h = propagate_sum(x, edge_index) + (1.0 + eps) * x
    return autograd.addmm(bias, h, weight) if bias is not None else autograd.matmul(h, weight)


def sage_conv(x, edge_index, weight_self, weight_neigh, bias=None):
    """GraphSAGE (mean aggregator): x @ W_self + mean_j x_j @ W_neigh + bias."""
    out = autograd.matmul(x, weight_self)
    return autograd.addmm(out if bias is None else out + bias, propagate_mean(x, edge_index), weight_neigh)
