"""CPU oracle for the gnn-ops-benchmark op hot path. TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package never does. It is a thin numpy/ctypes front end over
``gnnops_oracle.c`` (see that file's header for what is restated from where, and for what is and is
not pinned). Functions take the reference's argument order and meaning:

  scatter(src, index, dim, out=None, dim_size=None, reduce="sum")  ~ torch_scatter.scatter
      (reference call sites: op_bm_scripts/benchmark_scatter_add.py:18, benchmark_scatter_min.py:17, ...)
  index_select(input, dim, index)      ~ torch.index_select   (benchmark_native_index_select.py:14)
  index_add_(input, dim, index, src)   ~ Tensor.index_add_    (benchmark_native_index_add_.py:15)
  gather(input, dim, index)            ~ torch.gather         (benchmark_native_gather.py:16)

Arrays are numpy; bf16 data travels as ``np.uint16`` bit patterns tagged with ``dtype="bf16"``.
Pinning status: native ops are pinned to torch-CPU outputs in tests/golden; the torch_scatter family
is cross-checked against torch.scatter_reduce_ there, arg tie-breaking is PARITY UNPINNED.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

DT = {"f32": 0, "f16": 1, "bf16": 2}
REDUCE = {"sum": 0, "add": 0, "mean": 1, "min": 2, "max": 3, "mul": 4}
_NP_OF = {"f32": np.float32, "f16": np.float16, "bf16": np.uint16}


def build(force=False):
    """Compile gnnops_oracle.c with gcc (Makefile in this directory)."""
    src = os.path.join(_HERE, "gnnops_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        L.ora_scatter.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, ci, ci, ci, ci]
        L.ora_index_select.argtypes = [vp, vp, vp, i64, i64, i64, i64, ci]
        L.ora_gather.argtypes = [vp, vp, vp, i64, i64, i64, i64, ci]
        L.ora_plan.argtypes = [vp, i64, i64, vp, vp]
        L.ora_index_select_sum.argtypes = [vp, vp, vp, i64, i64, i64, i64, ci]
        L.ora_scatter_add_rows_f32.argtypes = [vp, vp, vp, i64, i64, i64]
        L.ora_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.ora_f16_to_f32.restype = ctypes.c_float
        L.ora_f32_to_f16.argtypes = [ctypes.c_float]
        L.ora_f32_to_f16.restype = ctypes.c_uint16
        L.ora_bf16_to_f32.argtypes = [ctypes.c_uint16]
        L.ora_bf16_to_f32.restype = ctypes.c_float
        L.ora_f32_to_bf16.argtypes = [ctypes.c_float]
        L.ora_f32_to_bf16.restype = ctypes.c_uint16
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _dtype_name(a, dtype):
    if dtype is not None:
        return dtype
    if a.dtype == np.float32:
        return "f32"
    if a.dtype == np.float16:
        return "f16"
    raise TypeError(f"pass dtype= for array dtype {a.dtype} (bf16 travels as uint16)")


def _bek(shape, dim):
    dim = dim % len(shape)
    B = int(np.prod(shape[:dim], dtype=np.int64))
    K = int(np.prod(shape[dim + 1:], dtype=np.int64))
    return B, int(shape[dim]), K, dim


def _check(rc, what):
    if rc == 1:
        raise IndexError(f"{what}: index out of range")
    if rc != 0:
        raise MemoryError(f"{what}: oracle failed with code {rc}")


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum", dtype=None):
    """torch_scatter.scatter semantics. Returns out, or (out, arg_out) for min/max."""
    src = np.ascontiguousarray(src)
    index = np.ascontiguousarray(index, dtype=np.int64)
    dt = _dtype_name(src, dtype)
    B, E, K, dim = _bek(src.shape, dim)
    if index.ndim == 1 and src.ndim >= 1 and index.shape[0] == E:
        full = 0
    elif index.shape == src.shape:
        full = 1
    else:  # torch_scatter broadcast(): expand a lower-rank index to src's shape
        idx = index
        if idx.ndim == 1:
            shape = [1] * src.ndim
            shape[dim] = -1
            idx = idx.reshape(shape)
        index = np.ascontiguousarray(np.broadcast_to(idx, src.shape))
        full = 1
    if out is not None:
        N = out.shape[dim]
        res = np.ascontiguousarray(out).copy()
        init = 1
    else:
        if dim_size is not None:
            N = int(dim_size)
        else:
            N = int(index.max()) + 1 if index.size else 0
        shape = list(src.shape)
        shape[dim] = N
        res = np.zeros(shape, dtype=_NP_OF[dt])
        init = 0
    r = REDUCE[reduce]
    arg = None
    if r in (2, 3):
        arg = np.empty(res.shape, dtype=np.int64)
    rc = lib().ora_scatter(_p(src), _p(index), _p(res), _p(arg), B, E, K, N, DT[dt], r, full, init)
    _check(rc, "scatter")
    return (res, arg) if arg is not None else res


def index_select(input, dim, index):
    input = np.ascontiguousarray(input)
    index = np.ascontiguousarray(index, dtype=np.int64)
    B, N, K, dim = _bek(input.shape, dim)
    E = index.shape[0]
    shape = list(input.shape)
    shape[dim] = E
    out = np.empty(shape, dtype=input.dtype)
    _check(lib().ora_index_select(_p(input), _p(index), _p(out), B, N, K, E, input.dtype.itemsize), "index_select")
    return out


def gather(input, dim, index):
    input = np.ascontiguousarray(input)
    index = np.ascontiguousarray(index, dtype=np.int64)
    B, N, K, dim = _bek(input.shape, dim)
    Bi, E, Ki, _ = _bek(index.shape, dim)
    if (Bi, Ki) != (B, K):
        raise ValueError("oracle.gather: index must match input outside `dim`")
    out = np.empty(index.shape, dtype=input.dtype)
    _check(lib().ora_gather(_p(input), _p(index), _p(out), B, N, K, E, input.dtype.itemsize), "gather")
    return out


def index_add_(input, dim, index, source, dtype=None):
    """In-place on a copy: returns input with source accumulated along dim at index (sequential order)."""
    return scatter(source, index, dim=dim, out=input, reduce="sum", dtype=dtype)


def plan(index, N):
    index = np.ascontiguousarray(index, dtype=np.int64)
    rowptr = np.empty(N + 1, dtype=np.int32)
    perm = np.empty(index.shape[0], dtype=np.int32)
    _check(lib().ora_plan(_p(index), index.shape[0], N, _p(rowptr), _p(perm)), "plan")
    return rowptr, perm


def index_select_sum(input, dim, index, dtype=None):
    input = np.ascontiguousarray(input)
    index = np.ascontiguousarray(index, dtype=np.int64)
    dt = _dtype_name(input, dtype)
    B, N, K, dim = _bek(input.shape, dim)
    out = ctypes.c_double(0.0)
    _check(lib().ora_index_select_sum(_p(input), _p(index), ctypes.byref(out), B, N, K, index.shape[0], DT[dt]),
           "index_select_sum")
    return out.value


def scatter_add_rows_f32(src, index, N):
    """The C port timed as bench.py's cpu_baseline (one core)."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    index = np.ascontiguousarray(index, dtype=np.int64)
    out = np.empty((N, src.shape[1]), dtype=np.float32)
    _check(lib().ora_scatter_add_rows_f32(_p(src), _p(index), _p(out), src.shape[0], src.shape[1], N), "scatter_add")
    return out


# ---- sparse / sort rows -------------------------------------------------------------------------------
def spmm(index, value, m, n, matrix, dtype=None):
    """torch_sparse.spmm(index [2,nnz], value [nnz] or None, m, n, matrix [n,D]) -> [m,D]."""
    index = np.ascontiguousarray(index, dtype=np.int64)
    matrix = np.ascontiguousarray(matrix)
    dt = _dtype_name(matrix, dtype)
    D = matrix.shape[1]
    out = np.empty((m, D), dtype=matrix.dtype)
    val = None if value is None else np.ascontiguousarray(value)
    row, col = np.ascontiguousarray(index[0]), np.ascontiguousarray(index[1])
    L = lib()
    L.ora_spmm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 4 + [ctypes.c_int]
    _check(L.ora_spmm(_p(row), _p(col), _p(val), _p(matrix), _p(out), index.shape[1], m, n, D, DT[dt]), "spmm")
    return out


def sort(input, dim=-1):
    """torch.sort(input, dim, stable=True) for float32: ascending, ties by position, NaN last, -0.0 == +0.0 as keys.
    Values are the ORIGINAL elements (values == input.gather(dim, indices) bit for bit: signed zeros and NaN payloads
    survive), as torch.sort returns them. numpy's stable argsort is the restatement."""
    x = np.ascontiguousarray(input, dtype=np.float32)
    idx = np.argsort(x, axis=dim, kind="stable").astype(np.int64)
    vals = np.take_along_axis(x, idx, axis=dim)
    return vals, idx


def coalesce(index, value, m, n, dtype=None):
    """torch_sparse.coalesce(index, value, m, n, op='add'): row-major sorted, duplicates summed in sorted order."""
    index = np.ascontiguousarray(index, dtype=np.int64)
    nnz = index.shape[1]
    key = index[0] * n + index[1]
    perm = np.argsort(key, kind="stable").astype(np.int64)
    skey = key[perm]
    head = np.ones(nnz, dtype=bool)
    head[1:] = skey[1:] != skey[:-1]
    seg_start = np.ascontiguousarray(np.nonzero(head)[0].astype(np.int64))
    ukey = skey[head]
    out_index = np.stack([ukey // n, ukey % n]).astype(np.int64) if nnz else np.zeros((2, 0), np.int64)
    if value is None:
        return out_index, None
    value = np.ascontiguousarray(value)
    dt = _dtype_name(value, dtype)
    C = int(value.size // nnz) if nnz else 0
    out = np.empty((len(seg_start),) + value.shape[1:], dtype=value.dtype)
    L = lib()
    L.ora_reduce_runs.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.c_void_p]
    _check(L.ora_reduce_runs(_p(value), _p(perm), _p(seg_start), len(seg_start), nnz, C, DT[dt], _p(out)), "coalesce")
    return out_index, out


def transpose_sparse(index, value, m, n, dtype=None):
    """torch_sparse.transpose(index, value, m, n): swap + coalesce over (n, m)."""
    index = np.ascontiguousarray(index, dtype=np.int64)
    return coalesce(np.stack([index[1], index[0]]), value, n, m, dtype=dtype)


def transpose_dense(mat):
    return np.ascontiguousarray(np.ascontiguousarray(mat).T)


def addmm(input, mat1, mat2, dtype=None):
    """float64 reference of torch.addmm(input, mat1, mat2) on the (rounded) 16-bit operands; input may be None."""
    mat1, mat2 = np.ascontiguousarray(mat1), np.ascontiguousarray(mat2)
    dt = _dtype_name(mat1, dtype)
    M, K = mat1.shape
    N = mat2.shape[1]
    out = np.empty((M, N), dtype=np.float64)
    inp = None if input is None else np.ascontiguousarray(input)
    L = lib()
    L.ora_addmm.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_int]
    _check(L.ora_addmm(_p(inp), _p(mat1), _p(mat2), _p(out), M, N, K, DT[dt]), "addmm")
    return out


def index_add_select_sum(input, dim, index, other, dtype=None):
    """float64 value of index_select(index_add(input, dim, index, other), dim, index).sum(dim)."""
    input, other = np.ascontiguousarray(input), np.ascontiguousarray(other)
    index = np.ascontiguousarray(index, dtype=np.int64)
    dt = _dtype_name(input, dtype)
    B, N, K, dim = _bek(input.shape, dim)
    E = other.shape[dim]
    shape = list(input.shape)
    del shape[dim]
    out = np.empty(shape, dtype=np.float64)
    L = lib()
    L.ora_index_add_select_sum.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 4 + [ctypes.c_int]
    _check(L.ora_index_add_select_sum(_p(input), _p(other), _p(index), _p(out), B, N, E, K, DT[dt]), "index_add_select_sum")
    return out


def spspmm(indexA, valueA, indexB, valueB, m, k, n, dtype=None):
    """torch_sparse.spspmm restated as expand - sort - compress: products enumerated over A's entries in order
    and, for each, over B's row in stored order; product rounded to the storage type; duplicates summed in that
    order by `coalesce` (reference call site: op_bm_scripts/benchmark_sparse_spspmm.py:12-14)."""
    indexA = np.ascontiguousarray(indexA, dtype=np.int64)
    indexB = np.ascontiguousarray(indexB, dtype=np.int64)
    valueA, valueB = np.ascontiguousarray(valueA), np.ascontiguousarray(valueB)
    dt = _dtype_name(valueA, dtype)
    rowptr, perm = plan(indexB[0], k)
    a32 = _widen(valueA, dt)
    b32 = _widen(valueB, dt)
    rows, cols, vals = [], [], []
    for a in range(indexA.shape[1]):
        kk = indexA[1, a]
        es = perm[rowptr[kk]:rowptr[kk + 1]]
        rows.append(np.full(len(es), indexA[0, a], dtype=np.int64))
        cols.append(indexB[1, es])
        vals.append((a32[a] * b32[es]).astype(np.float32))
    if rows:
        ex_index = np.stack([np.concatenate(rows), np.concatenate(cols)])
        ex_val = _narrow(np.concatenate(vals), dt)
    else:
        ex_index = np.zeros((2, 0), np.int64)
        ex_val = _narrow(np.zeros(0, np.float32), dt)
    return coalesce(ex_index, ex_val, m, n, dtype=dt)


def _widen(a, dt):
    if dt == "bf16":
        return (a.astype(np.uint32) << 16).view(np.float32)
    return a.astype(np.float32)


def _narrow(a32, dt):
    """fp32 -> storage type with round-to-nearest-even (through the C conversions for bf16)."""
    if dt == "f32":
        return a32.astype(np.float32)
    if dt == "f16":
        return a32.astype(np.float16)
    L = lib()
    return np.array([L.ora_f32_to_bf16(float(v)) for v in a32], dtype=np.uint16)


# ---- torch_scatter segment / composite ops -------------------------------------
def segment_csr(src, indptr, reduce="sum", dtype=None):
    """torch_scatter.segment_csr for a 1-D indptr along dim 0: scatter with index = segment id of each row."""
    indptr = np.asarray(indptr, dtype=np.int64)
    index = np.repeat(np.arange(len(indptr) - 1, dtype=np.int64), np.diff(indptr))
    return scatter(src, index, dim=0, dim_size=len(indptr) - 1, reduce=reduce, dtype=dtype)


def _groups(index, N):
    rowptr, perm = plan(index, N)
    return [perm[rowptr[n]:rowptr[n + 1]] for n in range(N)]


def composite(src, index, N, mode, eps=1e-12, unbiased=True):
    """scatter_softmax / log_softmax / logsumexp / std along dim 0 of a float32 [E, K] array, restating
    torch_scatter/composite/{softmax,logsumexp,std}.py (upstream 2.0.9; not in the reference tree): fp32,
    sequential over each group in source order. exp/log are numpy's, so device parity is to a few ulp."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    E, K = src.shape
    f32 = np.float32
    per_source = mode in ("softmax", "log_softmax")
    out = np.zeros((E, K) if per_source else (N, K), dtype=np.float32)
    for n, rows in enumerate(_groups(index, N)):
        x = src[rows]
        if mode == "std":
            s = np.zeros(K, f32)
            for r in x:
                s = s + r
            cnt = len(rows)
            mean = s / f32(max(cnt, 1))
            var = np.zeros(K, f32)
            for r in x:
                d = r - mean
                var = var + d * d
            c = max(cnt - 1, 1) if unbiased else max(cnt, 1)
            out[n] = np.sqrt(var / (f32(c) + f32(1e-6)))
            continue
        m = x.max(axis=0) if len(rows) else np.zeros(K, f32)
        s = np.zeros(K, f32)
        for r in x:
            s = s + np.exp(r - m, dtype=f32)
        if mode == "logsumexp":
            out[n] = m + np.log(s + f32(eps), dtype=f32)
        elif mode == "softmax":
            out[rows] = np.exp(x - m, dtype=f32) / s
        else:
            out[rows] = (x - m) - np.log(s + f32(eps), dtype=f32)
    return out
