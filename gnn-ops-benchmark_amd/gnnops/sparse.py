"""Sparse / sort front end: torch_sparse.{spmm, coalesce, transpose}, torch.sparse.mm(COO, dense),
Tensor.coalesce(), torch.sort and the dense transpose copy — reference call sites:
op_bm_scripts/benchmark_sparse_spmm.py:12-14, benchmark_sparse_coalesce.py:35-42,
benchmark_sparse_transpose.py:13-16, benchmark_native_sort.py:28-30.
"""
import torch

from . import _lib
from ._lib import check
from .ops import Plan, _check_index, _dtype_code, _norm_dim, _bek, _on, _require_gpu, _stream, get_plan


def _coo_rows_cols(index, what):
    _check_index(index, what)
    if index.dim() != 2 or index.size(0) != 2:
        raise ValueError(f"{what}: index must have shape [2, nnz]")
    index = index.contiguous()
    return index, index[0], index[1]


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def spmm(index, value, m, n, matrix):
    """torch_sparse.spmm(index, value, m, n, matrix): (m x n sparse) @ (n x D dense) -> m x D.
    Differentiable in value and matrix (gnnops/autograd.py)."""
    if _needs_grad(value, matrix):
        from . import autograd

        if matrix.dim() == 1:
            return autograd.spmm(index, value, m, n, matrix.unsqueeze(-1)).squeeze(-1)
        return autograd.spmm(index, value, m, n, matrix)
    return _spmm_raw(index, value, m, n, matrix)


def _spmm_raw(index, value, m, n, matrix):
    _require_gpu(index, value, matrix)
    index, row, col = _coo_rows_cols(index, "spmm")
    if matrix.dim() == 1:
        return _spmm_raw(index, value, m, n, matrix.unsqueeze(-1)).squeeze(-1)
    if matrix.dim() != 2:
        raise NotImplementedError("gnnops.spmm: matrix must be 1-D or 2-D")
    if matrix.size(0) != n:
        raise RuntimeError(f"spmm: matrix has {matrix.size(0)} rows, sparse operand has n={n} columns")
    dt = _dtype_code(matrix, "spmm")
    if value is not None and value.dtype != matrix.dtype:
        raise RuntimeError("spmm: value and matrix must have the same dtype")
    matrix = matrix.contiguous()
    plan = get_plan(row, m, owner=index, tag=0)  # COO -> CSR view (rowptr, perm), stable; cached under the [2, nnz] tensor
    return _spmm_launch(plan.rowptr, plan.perm, col, value, matrix, m, dt, plan, owner=index, tag=1)


def spmm_t(index, value, m, n, matrix):
    """(n x m)^T-side product without materialising the transposed index: out[i] = sum over entries (j, i) of
    value * matrix[j], i.e. rows of the result follow index[1] and the gathered rows follow index[0]. This is
    message passing over a PyG-style edge_index = (source, destination): out [n, D], matrix [m, D].
    Differentiable in value and matrix."""
    if _needs_grad(value, matrix):
        from . import autograd

        return autograd.spmm_t(index, value, m, n, matrix)
    return _spmm_t_raw(index, value, m, n, matrix)


def _spmm_t_raw(index, value, m, n, matrix):
    _require_gpu(index, value, matrix)
    index, src_rows, dst_rows = _coo_rows_cols(index, "spmm_t")
    if matrix.dim() != 2 or matrix.size(0) != m:
        raise RuntimeError("spmm_t: matrix must be [m, D]")
    dt = _dtype_code(matrix, "spmm_t")
    plan = get_plan(dst_rows, n, owner=index, tag=1)
    return _spmm_launch(plan.rowptr, plan.perm, src_rows, value, matrix.contiguous(), n, dt, plan, owner=index, tag=0)


def sddmm(rows_a, rows_b, a, b):
    """out[k] = <a[rows_a[k], :], b[rows_b[k], :]> (sampled dense-dense product; the value gradient of spmm)."""
    _require_gpu(rows_a, rows_b, a, b)
    _check_index(rows_a, "sddmm")
    _check_index(rows_b, "sddmm")
    if a.dim() != 2 or b.dim() != 2 or a.size(1) != b.size(1) or a.dtype != b.dtype:
        raise RuntimeError("sddmm: a and b must be 2-D with equal row length and dtype")
    if rows_a.numel() != rows_b.numel():
        raise RuntimeError("sddmm: rows_a and rows_b must have the same length")
    dt = _dtype_code(a, "sddmm")
    rows_a, rows_b, a, b = rows_a.contiguous(), rows_b.contiguous(), a.contiguous(), b.contiguous()
    out = torch.empty(rows_a.numel(), dtype=a.dtype, device=a.device)
    with _on(a.device):
        check(_lib.load().gnnops_sddmm(rows_a.data_ptr(), rows_b.data_ptr(), a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                       rows_a.numel(), a.size(1), dt, _stream()), "sddmm")
    return out


def spmm_csr(rowptr, col, value, matrix):
    """CSR x dense (BASELINE config 3's layout): rowptr int32/int64 [M+1], col int64 [nnz], value [nnz] or None."""
    if _needs_grad(value, matrix):
        raise NotImplementedError("gnnops.spmm_csr has no backward: use gnnops.spmm (COO) inside a training graph, or detach")
    _require_gpu(rowptr, col, value, matrix)
    _check_index(col, "spmm_csr")
    if rowptr.dtype == torch.int64:
        rowptr = rowptr.to(torch.int32)  # kernel ABI is int32 row pointers (nnz < 2^31)
    elif rowptr.dtype != torch.int32:
        raise RuntimeError("spmm_csr: rowptr must be int32 or int64")
    dt = _dtype_code(matrix, "spmm_csr")
    return _spmm_launch(rowptr.contiguous(), None, col.contiguous(), value, matrix.contiguous(), rowptr.numel() - 1, dt)


def _permute(t, perm, n):
    out = torch.empty(n, dtype=t.dtype, device=t.device)
    with _on(t.device):
        check(_lib.load().gnnops_permute(t.data_ptr(), perm.data_ptr(), out.data_ptr(), n, t.element_size(), _stream()), "permute")
    return out


def _csr_arrays(plan, col, value, owner=None, tag=0):
    """CSR column ids (and values) of a COO operand in plan order, kept on the plan.

    The entry is keyed on the tensor OBJECTS that own the data — ``owner`` is the [2, nnz] index tensor the caller
    passed (``col`` is a row view of it and a fresh object on every call, so it cannot be the key), ``tag`` says which of
    its rows is the column — plus their version counters; weak references guard against a recycled ``id``. Tensors
    without a version counter (created under torch.inference_mode()) are never cached."""
    from .ops import _version_of

    owner = col if owner is None else owner
    cacheable = _version_of(owner) is not None and (value is None or _version_of(value) is not None)
    key = None
    if cacheable:
        key = (id(owner), tag, owner._version, None if value is None else (id(value), value._version))
        cached = getattr(plan, "_csr", None)
        if cached is not None and cached[0] == key and cached[1]() is owner and (value is None or cached[2]() is value):
            return cached[3], cached[4]
    n = col.numel()
    col_csr = _permute(col, plan.perm, n)
    val_csr = _permute(value.contiguous(), plan.perm, n) if value is not None else None
    if cacheable:
        import weakref

        try:
            plan._csr = (key, weakref.ref(owner), weakref.ref(value) if value is not None else None, col_csr, val_csr)
        except AttributeError:
            pass
    return col_csr, val_csr


def _spmm_launch(rowptr, perm, col, value, matrix, m, dt, plan=None, owner=None, tag=0):
    if perm is not None and plan is not None and col.numel() > 0:
        col, value = _csr_arrays(plan, col, value, owner, tag)  # stream CSR arrays instead of chasing perm -> col per nonzero
        perm = None
    D = matrix.size(1)
    out = torch.empty((m, D), dtype=matrix.dtype, device=matrix.device)
    value_c = value.contiguous() if value is not None else None
    L = _lib.load()
    hub_bytes = L.gnnops_hub_workspace_bytes(col.numel(), D, 0)   # rows with more than 8192 nonzeros: csrc/hub.h
    hub_ws = torch.empty(hub_bytes, dtype=torch.uint8, device=matrix.device) if hub_bytes else None
    with _on(matrix.device):
        rc = L.gnnops_spmm_hubs(rowptr.data_ptr(), perm.data_ptr() if perm is not None else None, col.data_ptr(),
                                value_c.data_ptr() if value_c is not None else None, matrix.data_ptr(),
                                out.data_ptr(), m, D, col.numel(), matrix.size(0), dt,
                                hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes, _stream())
    check(rc, "spmm")
    return out


def sparse_mm(mat_a, mat_b):
    """torch.sparse.mm: sparse COO [m,n] x dense [n,D] -> dense (benchmark_sparse_spmm.py:12-14);
    sparse x sparse -> coalesced sparse COO (benchmark_sparse_spspmm.py:12-14)."""
    if not mat_a.is_sparse:
        raise RuntimeError("sparse_mm: first operand must be a sparse COO tensor")
    m, n = mat_a.shape
    if mat_b.is_sparse:
        idx, val = spspmm(mat_a._indices(), mat_a._values(), mat_b._indices(), mat_b._values(), m, n, mat_b.size(1))
        return torch.sparse_coo_tensor(idx, val, (m, mat_b.size(1)))._coalesced_(True)
    return spmm(mat_a._indices(), mat_a._values(), m, n, mat_b)


def spspmm(indexA, valueA, indexB, valueB, m, k, n, coalesced=False):
    """torch_sparse.spspmm(indexA, valueA, indexB, valueB, m, k, n): (m x k) @ (k x n), coalesced COO result.
    Two host round trips size the expansion and the result (the upstream op synchronises for the same reason)."""
    _require_gpu(indexA, valueA, indexB, valueB)
    indexA, rowA, colA = _coo_rows_cols(indexA, "spspmm")
    indexB, rowB, colB = _coo_rows_cols(indexB, "spspmm")
    dt = _dtype_code(valueA, "spspmm")
    if valueB.dtype != valueA.dtype:
        raise RuntimeError("spspmm: valueA and valueB must have the same dtype")
    valueA, valueB = valueA.contiguous(), valueB.contiguous()
    nnzA = indexA.size(1)
    dev = indexA.device
    L = _lib.load()
    planB = get_plan(rowB, k)
    ws_bytes = L.gnnops_spspmm_workspace_bytes(nnzA)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    with _on(dev):
        check(L.gnnops_spspmm_count(colA.data_ptr(), nnzA, planB.rowptr.data_ptr(), total.data_ptr(), ws.data_ptr(),
                                    ws_bytes, _stream()), "spspmm_count")
        P = int(total.item())
        if P >= 1 << 32:
            raise NotImplementedError(f"gnnops.spspmm: {P} partial products exceed the 2^32 limit")
        ex_index = torch.empty((2, P), dtype=torch.int64, device=dev)
        ex_value = torch.empty(P, dtype=valueA.dtype, device=dev)
        check(L.gnnops_spspmm_expand(rowA.data_ptr(), colA.data_ptr(), valueA.data_ptr(), nnzA, planB.rowptr.data_ptr(),
                                     planB.perm.data_ptr(), colB.data_ptr(), valueB.data_ptr(), ex_index[0].data_ptr(),
                                     ex_index[1].data_ptr(), ex_value.data_ptr(), dt, ws.data_ptr(), _stream()),
              "spspmm_expand")
    return coalesce(ex_index, ex_value, m, n)


def coalesce(index, value, m, n, op="add"):
    """torch_sparse.coalesce(index, value, m, n, op): row-major sorted, duplicates summed. Synchronises once
    to learn the number of distinct entries (it sizes the outputs), as the upstream op does."""
    if op not in ("add", "sum"):
        raise NotImplementedError(f"gnnops.coalesce: op={op!r} is not supported")
    _require_gpu(index, value)
    index, row, col = _coo_rows_cols(index, "coalesce")
    nnz = index.size(1)
    L = _lib.load()
    dev = index.device
    dt, C, value_c = 0, 0, None
    if value is not None:
        if value.size(0) != nnz:
            raise RuntimeError("coalesce: value.size(0) must equal index.size(1)")
        dt = _dtype_code(value, "coalesce")
        value_c = value.contiguous()
        C = value_c.numel() // max(nnz, 1) if nnz else 0
    out_index = torch.empty((2, nnz), dtype=torch.int64, device=dev)
    out_value = torch.empty_like(value_c) if value_c is not None else None
    count = torch.empty(1, dtype=torch.int64, device=dev)
    ws_bytes = L.gnnops_coalesce_workspace_bytes(nnz)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    with _on(dev):
        rc = L.gnnops_coalesce(row.data_ptr(), col.data_ptr(), value_c.data_ptr() if value_c is not None else None, nnz,
                               m, n, C, dt, out_index[0].data_ptr(), out_index[1].data_ptr(),
                               out_value.data_ptr() if out_value is not None else None, count.data_ptr(),
                               ws.data_ptr(), ws_bytes, _stream())
    check(rc, "coalesce")
    k = int(count.item())
    return out_index[:, :k], (out_value[:k] if out_value is not None else None)


def transpose(index, value, m, n, coalesced=True):
    """torch_sparse.transpose(index, value, m, n, coalesced): swap rows/cols; coalesce -> sorted by new row."""
    _require_gpu(index, value)
    index, row, col = _coo_rows_cols(index, "transpose")
    swapped = torch.stack([col, row], dim=0)
    if not coalesced:
        return swapped, value
    return coalesce(swapped, value, n, m)


def coalesce_sparse_tensor(mat):
    """Tensor.coalesce() for a sparse COO matrix (benchmark_sparse_coalesce.py:40-42)."""
    if not mat.is_sparse or mat.sparse_dim() != 2:
        raise NotImplementedError("gnnops.coalesce_sparse_tensor: 2-D sparse COO tensors only")
    idx, val = coalesce(mat._indices(), mat._values(), mat.size(0), mat.size(1))
    out = torch.sparse_coo_tensor(idx, val, mat.shape)
    return out._coalesced_(True)


def transpose_contiguous(mat):
    """torch.transpose(mat, 0, 1).contiguous() for a dense 2-D tensor (benchmark_sparse_transpose.py:13-16)."""
    _require_gpu(mat)
    if mat.dim() != 2:
        raise NotImplementedError("gnnops.transpose_contiguous: 2-D tensors only")
    eb = mat.element_size()
    if eb not in (1, 2, 4, 8):
        raise NotImplementedError(f"gnnops.transpose_contiguous: element size {eb}")
    mat = mat.contiguous()
    R, C = mat.shape
    out = torch.empty((C, R), dtype=mat.dtype, device=mat.device)
    with _on(mat.device):
        rc = _lib.load().gnnops_transpose2d(mat.data_ptr(), out.data_ptr(), R, C, eb, _stream())
    check(rc, "transpose2d")
    return out


def _transpose_batched(t3):
    """[B, R, C] -> [B, C, R] contiguous (any 1/2/4/8-byte dtype)."""
    Bn, R, C = t3.shape
    out = torch.empty((Bn, C, R), dtype=t3.dtype, device=t3.device)
    with _on(t3.device):
        check(_lib.load().gnnops_transpose_batched(t3.data_ptr(), out.data_ptr(), Bn, R, C, t3.element_size(), _stream()),
              "transpose_batched")
    return out


def _sort_rows(mat, descending, out_shape, idx32=False):
    """Rows of `mat` sorted on chip. idx32: the positions come back as int32 (they fit: rows are at most 40000 long) for a
    caller that widens them in a later pass of its own."""
    rows, E = mat.shape
    values = torch.empty(out_shape, dtype=torch.float32, device=mat.device)
    indices = torch.empty(out_shape, dtype=torch.int32 if idx32 else torch.int64, device=mat.device)
    L = _lib.load()
    one = L.gnnops_sort_rows_f32_i32 if idx32 else L.gnnops_sort_rows_f32
    two = L.gnnops_sort_rows2_f32_i32 if idx32 else L.gnnops_sort_rows2_f32
    with _on(mat.device):
        if E <= L.gnnops_sort_rows_max_len():
            check(one(mat.data_ptr(), values.data_ptr(), indices.data_ptr(), rows, E, 1 if descending else 0, _stream()), "sort_rows")
        else:  # up to twice the on-chip capacity: halves sorted on chip (positions as int32), then one rank merge
            tv = torch.empty((rows, E), dtype=torch.float32, device=mat.device)
            ti = torch.empty((rows, E), dtype=torch.int32, device=mat.device)
            check(two(mat.data_ptr(), values.data_ptr(), indices.data_ptr(), tv.data_ptr(), ti.data_ptr(),
                      rows, E, 1 if descending else 0, _stream()), "sort_rows2")
    return values, indices


_SORT_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.int32: 3, torch.int64: 4, torch.float64: 5}


def sort(input, dim=-1, descending=False, stable=False):
    """torch.sort(input, dim, descending, stable): (values, indices int64), always stable.
    float32 / float16 / bfloat16 / int32 along any dim; int64 / float64 for 1-D tensors."""
    _require_gpu(input)
    if input.dtype not in _SORT_DT:
        raise NotImplementedError(f"gnnops.sort: dtype {input.dtype} is not supported")
    if input.dim() == 0:
        return input.clone(), torch.zeros((), dtype=torch.int64, device=input.device)
    dim = _norm_dim(dim, input.dim(), "sort")
    input = input.contiguous()
    B, E, K = _bek(input.shape, dim)
    sd = _SORT_DT[input.dtype]
    if sd >= 4 and B * K != 1:
        raise NotImplementedError(f"gnnops.sort: {input.dtype} is sorted for 1-D tensors only")
    L = _lib.load()
    if input.dtype == torch.float32 and 32 <= E <= L.gnnops_sort_rows2_max_len() and B * K > 1:
        # rows that fit in LDS are sorted on chip; along dim 0 of a matrix via our tiled transposes
        if K == 1:
            return _sort_rows(input.view(B, E), descending, input.shape)
        # [B, E, K] -> [B, K, E] by our tiled transposes, sort the B*K rows, transpose both results back
        if B == 1:
            # one matrix: the positions stay int32 until the transpose back, which widens them on the way (3.2 GB less written
            # and 3.2 GB less read at (28200, 28200))
            v, i32 = _sort_rows(transpose_contiguous(input.view(E, K)), descending, (K, E), idx32=True)
            idx = torch.empty(input.shape, dtype=torch.int64, device=input.device)
            with _on(input.device):
                check(L.gnnops_transpose2d_cvt(i32.data_ptr(), idx.data_ptr(), K, E, 1, _stream()), "transpose2d_cvt")
            return transpose_contiguous(v).view(input.shape), idx
        v, i = _sort_rows(_transpose_batched(input.view(B, E, K)).view(B * K, E), descending, (B, K, E))
        return _transpose_batched(v).view(input.shape), _transpose_batched(i).view(input.shape)
    values = torch.empty_like(input)
    indices = torch.empty(input.shape, dtype=torch.int64, device=input.device)
    ws_bytes = L.gnnops_sort_workspace_bytes(B, E, K, sd)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=input.device)
    with _on(input.device):
        rc = L.gnnops_sort(input.data_ptr(), values.data_ptr(), indices.data_ptr(), B, E, K, sd, 1 if descending else 0,
                           ws.data_ptr(), ws_bytes, _stream())
    check(rc, "sort")
    return values, indices
