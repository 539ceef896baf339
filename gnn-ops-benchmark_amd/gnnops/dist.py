"""Destination-partitioned scatter across the GPUs of one node (BASELINE config 5).

One process per GPU. Edges are partitioned by position (rank g holds src_g [E_g, D] and index_g [E_g]
with GLOBAL destination ids anywhere in [0, n_total)); destination rows are partitioned contiguously
(rank g owns rows [g*n_total/G, (g+1)*n_total/G)). The path has exactly one exchange step:

  1. local segment reduce into a partial [n_total, D] buffer (our kernels, rows grouped by owner),
  2. ONE reduce-scatter (RCCL over xGMI; `torch.distributed` backend "nccl" is RCCL on ROCm) that
     leaves each rank with the summed slab it owns.

The reference has no distributed code; this module is the MI355X design for config 5.
`local_scatter` is injectable so the exchange logic is testable on CPU with the gloo backend (tests
pass the oracle there); the product default is the HIP op and refuses CPU tensors like everything else.
"""
import torch
import torch.distributed as dist

_REDUCE_OP = {"sum": dist.ReduceOp.SUM, "add": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}


def owned_rows(n_total, rank, world):
    """Row range [lo, hi) of the destination slab rank `rank` owns."""
    if n_total % world != 0:
        raise ValueError(f"n_total={n_total} must be divisible by world size {world}")
    per = n_total // world
    return rank * per, (rank + 1) * per


def sharded_scatter(src_local, index_local, n_total, reduce="sum", group=None, local_scatter=None, out_slab=None):
    """Scatter-reduce this rank's edges into global destinations and return the slab this rank owns.

    sum / min / max use one reduce-scatter; mean = reduce-scatter of sums and of counts, then divide.
    (arg_out across ranks needs a (value, index) pair reduction — SURVEY.md §8f — and is not provided.)
    """
    if local_scatter is None:
        from .ops import scatter as local_scatter
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = owned_rows(n_total, rank, world)
    if reduce == "mean":
        sums = sharded_scatter(src_local, index_local, n_total, "sum", group, local_scatter)
        ones = torch.ones((src_local.shape[0], 1), dtype=src_local.dtype, device=src_local.device)
        cnt = sharded_scatter(ones, index_local, n_total, "sum", group, local_scatter)
        return sums / cnt.clamp_(min=1)
    if reduce not in _REDUCE_OP:
        raise ValueError(f"sharded_scatter: reduce {reduce!r} has no single-collective form")
    partial = local_scatter(src_local, index_local, 0, None, n_total, reduce)
    if isinstance(partial, tuple):
        partial = partial[0]
    if reduce in ("min", "max"):
        # local empties were zero-filled (torch_scatter convention); make them neutral for the collective
        cnt = local_scatter(torch.ones((src_local.shape[0], 1), dtype=src_local.dtype, device=src_local.device),
                            index_local, 0, None, n_total, "sum")
        neutral = float("inf") if reduce == "min" else float("-inf")
        partial = torch.where(cnt > 0, partial, torch.full_like(partial, neutral))
    if out_slab is None:
        out_slab = torch.empty((hi - lo,) + tuple(partial.shape[1:]), dtype=partial.dtype, device=partial.device)
    dist.reduce_scatter_tensor(out_slab, partial.contiguous(), op=_REDUCE_OP[reduce], group=group)
    if reduce in ("min", "max"):
        out_slab = torch.where(torch.isinf(out_slab), torch.zeros_like(out_slab), out_slab)
    return out_slab
