"""torch_scatter's segment and composite ops on the gfx950 kernels:
segment_csr / segment_coo / gather_csr / gather_coo, scatter_softmax / scatter_log_softmax /
scatter_logsumexp / scatter_std. These are what PyG's MessagePassing.aggregate, global pooling and attention
softmax call on the reference's OpProfiler path (graph_benchmark/models/ptg_models.py:62-78,176-195,238-258).
"""
import torch

from . import _lib
from ._lib import REDUCE_CODE, check
from .ops import Plan, _bek, _check_index, _dtype_code, _norm_dim, _on, _require_gpu, _row_index_of, _stream, get_plan

_MODES = {"softmax": 0, "log_softmax": 1, "logsumexp": 2, "std": 3}


class _CSR:
    """rowptr-only plan: segments are contiguous runs of src (perm = identity)."""

    __slots__ = ("rowptr", "perm", "E", "N")

    def __init__(self, rowptr, E):
        self.rowptr, self.perm, self.E, self.N = rowptr, None, E, rowptr.numel() - 1


def _as_int32_rowptr(indptr):
    if indptr.dtype == torch.int32:
        return indptr.contiguous()
    if indptr.dtype == torch.int64:
        return indptr.to(torch.int32)
    raise RuntimeError("indptr must be int32 or int64")


def _reduce_over(plan, src, dim, reduce, want_arg):
    rcode = REDUCE_CODE[reduce]
    dt = _dtype_code(src, "segment")
    B, E, K = _bek(src.shape, dim)
    shape = list(src.shape)
    shape[dim] = plan.N
    out = torch.empty(shape, dtype=src.dtype, device=src.device)
    arg = torch.empty(shape, dtype=torch.int64, device=src.device) if want_arg else None
    with _on(src.device):
        rc = _lib.load().gnnops_segment_reduce(src.data_ptr(), plan.rowptr.data_ptr(),
                                               plan.perm.data_ptr() if plan.perm is not None else None, out.data_ptr(),
                                               arg.data_ptr() if arg is not None else None, B, E, K, plan.N, dt, rcode, 0,
                                               _stream())
    check(rc, "segment_reduce")
    return (out, arg) if want_arg else out


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _segment_csr_raw(src, indptr, reduce="sum"):
    _require_gpu(src, indptr)
    if indptr.dim() != 1:
        raise NotImplementedError("gnnops.segment_csr: indptr must be 1-D")
    src = src.contiguous()
    plan = _CSR(_as_int32_rowptr(indptr), src.size(0))
    return _reduce_over(plan, src, 0, reduce, reduce in ("min", "max"))


def segment_csr(src, indptr, out=None, reduce="sum"):
    """torch_scatter.segment_csr(src, indptr, out, reduce) for a 1-D indptr: segments along dim 0... the last
    dimension of indptr, i.e. dim = indptr.dim() - 1 = 0. Differentiable in src (gnnops/autograd.py)."""
    if out is not None:
        raise NotImplementedError("gnnops.segment_csr: out= is not supported")
    if _needs_grad(src):
        from . import autograd

        return autograd.segment_csr(src, indptr, reduce)
    return _segment_csr_raw(src, indptr, reduce)


def expand_rowptr(indptr, E):
    """int64 [E]: the segment of the CSR pointer that holds each position (N = indptr.numel() - 1 where none does)."""
    _require_gpu(indptr)
    rowptr = _as_int32_rowptr(indptr)
    index = torch.empty(E, dtype=torch.int64, device=indptr.device)
    with _on(indptr.device):
        check(_lib.load().gnnops_rowptr_expand(rowptr.data_ptr(), rowptr.numel() - 1, E, index.data_ptr(), _stream()),
              "rowptr_expand")
    return index


def rowptr_from_sorted(index, N):
    """int32 CSR pointer of a sorted int64 index with values in [0, N)."""
    _require_gpu(index)
    _check_index(index, "rowptr_from_sorted")
    index = index.contiguous()
    L = _lib.load()
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=index.device)
    ws_bytes = L.gnnops_rowptr_workspace_bytes(N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=index.device)
    with _on(index.device):
        check(L.gnnops_rowptr_from_sorted(index.data_ptr(), index.numel(), N, rowptr.data_ptr(), ws.data_ptr(), ws_bytes,
                                          _stream()), "rowptr_from_sorted")
    return rowptr


def segment_coo(src, index, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.segment_coo(src, index, out, dim_size, reduce): index 1-D, SORTED, along dim 0.
    Differentiable in src (gnnops/autograd.py)."""
    if out is not None:
        raise NotImplementedError("gnnops.segment_coo: out= is not supported")
    if _needs_grad(src):
        from . import autograd

        return autograd.segment_coo(src, index, dim_size, reduce)
    return _segment_coo_raw(src, index, dim_size, reduce)


def _segment_coo_raw(src, index, dim_size=None, reduce="sum"):
    _require_gpu(src, index)
    if index.dim() != 1:
        raise NotImplementedError("gnnops.segment_coo: index must be 1-D")
    from .ops import index_max

    N = int(dim_size) if dim_size is not None else (index_max(index) + 1 if index.numel() else 0)
    src = src.contiguous()
    plan = _CSR(rowptr_from_sorted(index, N), src.size(0))
    return _reduce_over(plan, src, 0, reduce, reduce in ("min", "max"))


def gather_csr(src, indptr, out=None):
    """torch_scatter.gather_csr: out[e] = src[segment containing e], e in [0, indptr[-1]) — reading indptr[-1] back
    sizes the output, as in the upstream op. Differentiable in src."""
    if out is not None:
        raise NotImplementedError("gnnops.gather_csr: out= is not supported")
    _require_gpu(src, indptr)
    if indptr.dim() != 1:
        raise NotImplementedError("gnnops.gather_csr: indptr must be 1-D")
    E = int(indptr[-1].item()) if indptr.numel() else 0
    if _needs_grad(src):
        from . import autograd

        return autograd.gather_csr(src, indptr, E)
    from .ops import index_select

    # positions below indptr[0] belong to no segment (index == N): they read an appended zero row
    index = expand_rowptr(indptr, E)
    if int(indptr[0].item()) > 0:
        src = torch.cat([src, src.new_zeros((1,) + tuple(src.shape[1:]))])
    return index_select(src, 0, index)


def gather_coo(src, index, out=None):
    """torch_scatter.gather_coo: out[e] = src[index[e]] along dim 0. Differentiable in src."""
    if out is not None:
        raise NotImplementedError("gnnops.gather_coo: out= is not supported")
    if _needs_grad(src):
        from . import autograd

        return autograd.gather_coo(src, index)
    from .ops import index_select

    return index_select(src, 0, index)


def _composite(src, index, dim, dim_size, mode, param):
    is_plan = isinstance(index, Plan)
    _require_gpu(src, None if is_plan else index)
    dt = _dtype_code(src, "scatter_" + mode)
    dim = _norm_dim(dim, src.dim(), "scatter_" + mode)
    src = src.contiguous()
    B, E, K = _bek(src.shape, dim)
    if is_plan:
        plan = index
    else:
        _check_index(index, "scatter_" + mode)
        row = _row_index_of(index, src, dim)
        if row is None:
            raise NotImplementedError(f"gnnops.scatter_{mode}: a per-element index is not supported (row index only)")
        from .ops import index_max

        N = int(dim_size) if dim_size is not None else (index_max(row) + 1 if row.numel() else 0)
        plan = get_plan(row, N)
    per_source = mode in ("softmax", "log_softmax")
    shape = list(src.shape)
    if not per_source:
        shape[dim] = plan.N
    out = torch.empty(shape, dtype=src.dtype, device=src.device)
    L = _lib.load()
    hub_bytes = L.gnnops_hub_workspace_bytes(E, K, _lib.MIN) if B == 1 else 0   # groups with more than 8192 members: hub.h
    hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=src.device) if hub_bytes else None
    with _on(src.device):
        rc = L.gnnops_segment_composite_hubs(src.data_ptr(), plan.rowptr.data_ptr(),
                                             plan.perm.data_ptr() if plan.perm is not None else None, out.data_ptr(),
                                             B, E, K, plan.N, dt, _MODES[mode], float(param),
                                             hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes, _stream())
    check(rc, "segment_composite")
    return out


def _maybe_grad(src, index, dim, dim_size, mode, eps):
    if torch.is_grad_enabled() and src.requires_grad and not isinstance(index, Plan):
        from . import autograd

        return autograd.composite(src, index, dim, dim_size, mode, eps)
    return _composite(src, index, dim, dim_size, mode, eps)


def scatter_softmax(src, index, dim=-1, dim_size=None):
    return _maybe_grad(src, index, dim, dim_size, "softmax", 0.0)


def scatter_log_softmax(src, index, dim=-1, eps=1e-12, dim_size=None):
    return _maybe_grad(src, index, dim, dim_size, "log_softmax", eps)


def scatter_logsumexp(src, index, dim=-1, out=None, dim_size=None, eps=1e-12):
    if out is not None:
        raise NotImplementedError("gnnops.scatter_logsumexp: out= is not supported")
    return _maybe_grad(src, index, dim, dim_size, "logsumexp", eps)


def scatter_std(src, index, dim=-1, out=None, dim_size=None, unbiased=True):
    if out is not None:
        raise NotImplementedError("gnnops.scatter_std: out= is not supported")
    return _maybe_grad(src, index, dim, dim_size, "std", 1.0 if unbiased else 0.0)
