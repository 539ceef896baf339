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
